// transcript.hpp -- device-resident Fiat-Shamir transcript for gfx950.
//
// Replaces transcripts/fiat-shamir/src/fiat_shamir.rs:10-40 (SHA-256 hash chain:
// commit = update; challenge = finalize, reset, re-seed with the digest;
// evaluate_challenge_into_field = from_be_bytes_mod_order) so that the sumcheck
// round loop never leaves the GPU between rounds.  One lane runs it; the state
// lives in global memory between kernels.
#pragma once
#include "fp.hpp"

namespace zk {

struct Sha256State {
    uint32_t h[8];
    uint32_t buf[16];   // pending block, big-endian words
    uint32_t fill;      // bytes pending in buf (always a multiple of 4 here: we only absorb 32-byte items)
    uint32_t pad_;
    uint64_t len;       // total bytes absorbed
};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_rotateright32(x, n); }

static __constant__ uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// One compression.  Kept out of line and rolled (16 rounds + 3 x 16 rounds with the message schedule):
// the transcript runs on a single wave, where instruction fetch of a fully unrolled body costs more
// than the loop.  v_bitop3_b32 folds each 3-input boolean (xor3 / choose / majority) into one instruction.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t choose(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ uint32_t majority(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

// A lone wave issues one instruction per 4 (VOP2) or 5 (VOP3) cycles whatever the dependencies (profiles/r02/ubench_salu_gfx950.txt),
// so a round costs its instruction count: 6 rotates + 4 three-input booleans + 4 additions -- two of them v_add3_u32, which the
// compiler does not form by itself here (it emitted six two-input additions).
__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define ZK_SHA_ROUND(a, b, c, d, e, f, g, h, kw)                                      \
    {                                                                                 \
        uint32_t t1 = add3(h + (kw), xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25)), choose(e, f, g)); \
        d += t1;                                                                      \
        h = add3(t1, xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22)), majority(a, b, c));             \
    }
#define ZK_SHA_8ROUNDS(W, KB)                                    \
    ZK_SHA_ROUND(a, b, c, d, e, f, g, hh, SHA256_K[KB + 0] + W[0]) \
    ZK_SHA_ROUND(hh, a, b, c, d, e, f, g, SHA256_K[KB + 1] + W[1]) \
    ZK_SHA_ROUND(g, hh, a, b, c, d, e, f, SHA256_K[KB + 2] + W[2]) \
    ZK_SHA_ROUND(f, g, hh, a, b, c, d, e, SHA256_K[KB + 3] + W[3]) \
    ZK_SHA_ROUND(e, f, g, hh, a, b, c, d, SHA256_K[KB + 4] + W[4]) \
    ZK_SHA_ROUND(d, e, f, g, hh, a, b, c, SHA256_K[KB + 5] + W[5]) \
    ZK_SHA_ROUND(c, d, e, f, g, hh, a, b, SHA256_K[KB + 6] + W[6]) \
    ZK_SHA_ROUND(b, c, d, e, f, g, hh, a, SHA256_K[KB + 7] + W[7])

__device__ __noinline__ void sha256_compress(uint32_t (&h)[8], const uint32_t (&blk)[16]) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = blk[i];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    {
        const uint32_t* W = w;
        ZK_SHA_8ROUNDS(W, 0)
        W = w + 8;
        ZK_SHA_8ROUNDS(W, 8)
    }
#pragma unroll 1
    for (int base = 16; base < 64; base += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t w15 = w[(j + 1) & 15], w2 = w[(j + 14) & 15];
            uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            w[j] = w[j] + s0 + w[(j + 9) & 15] + s1;
        }
        const uint32_t* W = w;
        ZK_SHA_8ROUNDS(W, base)
        W = w + 8;
        ZK_SHA_8ROUNDS(W, base + 8)
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// ---- split compression: the message schedule on another wave ---------------------------------------------------
// A compression is 64 rounds of ~14 instructions on the working state plus 48 schedule steps of ~10 -- and the
// schedule does not depend on the state.  In the serial prover kernel a helper wave computes kw[t] = K[t] + W[t] into
// LDS while the transcript wave runs the rounds; chunks of 16 words are handed over through a counter that only
// grows (round r uses the values 4 r + 1 .. 4 r + 4), written after the data by the same wave (LDS operations of
// one wave complete in order).
__device__ __forceinline__ void sha256_wait_flag(volatile uint32_t* flag, uint32_t want) {
    while (*flag < want) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// producer: w[0..15] = the block; publishes kw[first_chunk*16 .. 63] chunk by chunk
__device__ __forceinline__ void sha256_schedule_to_lds(uint32_t (&w)[16], uint32_t* __restrict__ kw, volatile uint32_t* flag,
                                                    uint32_t flag_base, uint32_t first_chunk) {
    const bool writer = (threadIdx.x & 63) == 0;
    if (first_chunk == 0) {
        if (writer) {
#pragma unroll
            for (int j = 0; j < 16; ++j) kw[j] = SHA256_K[j] + w[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the data before the counter
            *flag = flag_base + 1;
        }
    }
#pragma unroll 1
    for (int base = 16; base < 64; base += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t w15 = w[(j + 1) & 15], w2 = w[(j + 14) & 15];
            uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            w[j] = w[j] + s0 + w[(j + 9) & 15] + s1;
        }
        if (writer) {
#pragma unroll
            for (int j = 0; j < 16; ++j) kw[base + j] = SHA256_K[base + j] + w[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            *flag = flag_base + 1 + base / 16;
        }
    }
}
#define ZK_SHA_8ROUNDS_KW(KW)                                  \
    ZK_SHA_ROUND(a, b, c, d, e, f, g, hh, KW[0])               \
    ZK_SHA_ROUND(hh, a, b, c, d, e, f, g, KW[1])               \
    ZK_SHA_ROUND(g, hh, a, b, c, d, e, f, KW[2])               \
    ZK_SHA_ROUND(f, g, hh, a, b, c, d, e, KW[3])               \
    ZK_SHA_ROUND(e, f, g, hh, a, b, c, d, KW[4])               \
    ZK_SHA_ROUND(d, e, f, g, hh, a, b, c, KW[5])               \
    ZK_SHA_ROUND(c, d, e, f, g, hh, a, b, KW[6])               \
    ZK_SHA_ROUND(b, c, d, e, f, g, hh, a, KW[7])
// consumer: rounds 0..15 from `blk` when given (else from kw chunk 0), rounds 16..63 from the kw chunks as they arrive
__device__ __forceinline__ void sha256_compress_kw(uint32_t (&h)[8], const uint32_t* blk /* 16 words or nullptr */,
                                                const uint32_t* __restrict__ kw, volatile uint32_t* flag, uint32_t flag_base) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    uint32_t v[16];
    if (blk) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = SHA256_K[j] + blk[j];
    } else {
        sha256_wait_flag(flag, flag_base + 1);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = kw[j];
    }
    {
        const uint32_t* KW = v;
        ZK_SHA_8ROUNDS_KW(KW)
        KW = v + 8;
        ZK_SHA_8ROUNDS_KW(KW)
    }
#pragma unroll 1
    for (int base = 16; base < 64; base += 16) {
        sha256_wait_flag(flag, flag_base + 1 + base / 16);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = kw[base + j];
        const uint32_t* KW = v;
        ZK_SHA_8ROUNDS_KW(KW)
        KW = v + 8;
        ZK_SHA_8ROUNDS_KW(KW)
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// whole-block forms for the composed provers' round closing: every block of the padded message is known before the
// hash starts, so all schedules are computed side by side (one wave per block) and the hash wave runs state rounds only
__device__ __forceinline__ void sha256_schedule_block(const uint32_t* __restrict__ blk, uint32_t* __restrict__ kw) {
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = blk[j];
    const bool writer = (threadIdx.x & 63) == 0;
    if (writer) {
#pragma unroll
        for (int j = 0; j < 16; ++j) kw[j] = SHA256_K[j] + w[j];
    }
#pragma unroll 1
    for (int base = 16; base < 64; base += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t w15 = w[(j + 1) & 15], w2 = w[(j + 14) & 15];
            uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
            uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
            w[j] = w[j] + s0 + w[(j + 9) & 15] + s1;
        }
        if (writer) {
#pragma unroll
            for (int j = 0; j < 16; ++j) kw[base + j] = SHA256_K[base + j] + w[j];
        }
    }
}
__device__ __forceinline__ void sha256_rounds_block(uint32_t (&h)[8], const uint32_t* __restrict__ kw) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll 1
    for (int base = 0; base < 64; base += 16) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = kw[base + j];
        const uint32_t* KW = v;
        ZK_SHA_8ROUNDS_KW(KW)
        KW = v + 8;
        ZK_SHA_8ROUNDS_KW(KW)
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// Out-of-line Montgomery product for the single-wave control paths (keeps those kernels small).
__device__ __noinline__ Fr fr_mul_outlined(Fr a, Fr b) { return a * b; }   // by value: arguments travel in registers, not through scratch
// Montgomery form -> canonical integer (into_bigint): the reduction half of a product only (x * 1 has no
// multiplication part): 8 words of word-serial REDC, 64 mads instead of 128.
__device__ __noinline__ Fr fr_from_mont_outlined(Fr a) {
    uint32_t x[9];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a.l[i];
    x[8] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t m = FrParams::mul_inv(x[0]);
        uint64_t c = ((uint64_t)m * FrParams::p(0) + x[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            uint64_t s = (uint64_t)m * FrParams::p(j) + x[j] + c;
            x[j - 1] = (uint32_t)s;
            c = s >> 32;
        }
        uint64_t s = (uint64_t)x[8] + c;
        x[7] = (uint32_t)s;
        x[8] = (uint32_t)(s >> 32);
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = x[i];
    r.reduce_once();   // a < r  =>  result < r already; kept for inputs up to 2r
    return r;
}
__device__ __forceinline__ Fr fr_to_mont_outlined(const Fr& a) {
    Fr r2;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) r2.l[i] = FrParams::r2(i);
    return fr_mul_outlined(a, r2);
}

// Register-resident transcript (loaded from / stored to a Sha256State in global memory).
struct Transcript {
    uint32_t h[8];
    uint32_t buf[16];
    uint32_t fill_words;   // words pending
    uint64_t len;

    __device__ __forceinline__ void init() {   // FiatShamirTranscript::new  fiat_shamir.rs:11-15
        h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
        h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
        fill_words = 0;
        len = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) buf[i] = 0;
    }
    __device__ __forceinline__ void load(const Sha256State* s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = s->h[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) buf[i] = s->buf[i];
        fill_words = s->fill >> 2;
        len = s->len;
    }
    __device__ __forceinline__ void store(Sha256State* s) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) s->h[i] = h[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) s->buf[i] = buf[i];
        s->fill = fill_words << 2;
        s->len = len;
    }
    // absorb 8 big-endian words (32 bytes).  fill_words stays a multiple of 8, so the
    // dynamic index below only takes the values 0 and 8.
    __device__ __forceinline__ void commit_words8(const uint32_t (&wds)[8]) {   // commit  fiat_shamir.rs:17-19
        if (fill_words == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) buf[i] = wds[i];
            fill_words = 8;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) buf[8 + i] = wds[i];
            sha256_compress(h, buf);
            fill_words = 0;
        }
        len += 32;
    }
    // to_bytes_be() of an element already converted to its canonical integer (into_bigint), absorbed
    __device__ __forceinline__ void commit_canonical(const Fr& c) {
        uint32_t wds[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wds[i] = c.l[7 - i];
        commit_words8(wds);
    }
    // into_bigint().to_bytes_be() of a Montgomery-form element, absorbed (sumcheck/src/utils.rs:7-9)
    __device__ __forceinline__ void commit_fr(const Fr& v_mont) {
        Fr c = fr_from_mont_outlined(v_mont);
        uint32_t wds[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wds[i] = c.l[7 - i];   // big-endian byte string == words MSW first
        commit_words8(wds);
    }
    // challenge(): finalize_reset + update(digest)   fiat_shamir.rs:21-25
    __device__ __forceinline__ void challenge(uint32_t (&digest)[8]) {
        uint64_t bits = len * 8;
        if (fill_words == 0) {
            buf[0] = 0x80000000u;
#pragma unroll
            for (int i = 1; i < 14; ++i) buf[i] = 0;
        } else {   // 8 words pending
            buf[8] = 0x80000000u;
#pragma unroll
            for (int i = 9; i < 14; ++i) buf[i] = 0;
        }
        buf[14] = (uint32_t)(bits >> 32);
        buf[15] = (uint32_t)bits;
        sha256_compress(h, buf);
#pragma unroll
        for (int i = 0; i < 8; ++i) digest[i] = h[i];
        init();
        commit_words8(digest);
    }
    // evaluate_challenge_into_field: from_be_bytes_mod_order(digest)  fiat_shamir.rs:27-29.
    // digest < 2^256 < 3r, so at most two subtractions of r bring it into range.  Returns Montgomery form.
    // canonical integer of the challenge (digest mod r); the Montgomery form is one more product away
    __device__ __forceinline__ Fr challenge_canonical() {
        uint32_t d[8];
        challenge(d);
        Fr v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v.l[i] = d[7 - i];
        v.reduce_once();
        v.reduce_once();
        return v;
    }
    __device__ __forceinline__ Fr challenge_fr() { return fr_to_mont_outlined(challenge_canonical()); }
};

}  // namespace zk

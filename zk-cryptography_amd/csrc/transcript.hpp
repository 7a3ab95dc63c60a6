// transcript.hpp -- device-resident Fiat-Shamir transcript for gfx950.
//
// Replaces transcripts/fiat-shamir/src/fiat_shamir.rs:10-40 (SHA-256 hash chain:
// commit = update; challenge = finalize, reset, re-seed with the digest;
// evaluate_challenge_into_field = from_be_bytes_mod_order) so that the sumcheck
// round loop never leaves the GPU between rounds.  One WAVE runs it -- the state
// rounds on six of its lanes, the message schedule on sixteen (below) -- and the state
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

// ---- message schedule on the sixteen lanes of a row -------------------------------------------------------------
// The serial forms above spend ~12 instructions per schedule word on a wave that pays per instruction: 1.75 us per block, more
// than the six-lane state rounds below need for it.  Here lane i of a row of 16 holds word i of a chunk of 16, and
//     w[t] = w[t-16] + s0(w[t-15]) + w[t-7] + s1(w[t-2])
// is evaluated for the whole chunk at once: the terms that come from the previous chunk (w[t-16]; s0(w[t-15]) for i < 15; w[t-7] for
// i < 7; s1(w[t-2]) for i < 2) with three DPP additions, the others by iteration -- after step k lanes 0 .. 2k+1 hold their final
// words (s1 reaches two lanes back, w[t-7] seven: both are final by then), so seven steps of six instructions finish the chunk;
// lane 15's s0(w[t-15]) is s0 of the new chunk's word 0, final after the first step and added once.  ~65 instructions per chunk
// instead of ~200, and the four rows of a wave take four blocks side by side.
template <int CTRL> __device__ __forceinline__ uint32_t dpp_row0(uint32_t v) {     // lanes without a source read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// c.l[j] with a per-lane j in 0..7 (a message word of this lane out of an element every lane holds)
__device__ __forceinline__ uint32_t fr_limb_by_lane(const Fr& c, uint32_t j) {
    uint32_t v = c.l[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) v = (j == (uint32_t)k) ? c.l[k] : v;
    return v;
}
__device__ __forceinline__ uint32_t sha_s0(uint32_t x) { return xor3(rotr32(x, 7), rotr32(x, 18), x >> 3); }
__device__ __forceinline__ uint32_t sha_s1(uint32_t x) { return xor3(rotr32(x, 17), rotr32(x, 19), x >> 10); }
// w: this lane's word of the previous chunk -> its word of the next one.  Every lane of the wave runs it (rows are independent).
__device__ __forceinline__ uint32_t sha256_schedule_chunk_rows(uint32_t wp) {
    constexpr int SHL = 0x100, SHR = 0x110, ROR = 0x120;
    const uint32_t lane15 = ((threadIdx.x & 15) == 15) ? 0xFFFFFFFFu : 0u;
    uint32_t base = wp + dpp_row0<SHL + 1>(sha_s0(wp));
    base += dpp_row0<SHL + 9>(wp);
    base += dpp_row0<SHL + 14>(sha_s1(wp));
    uint32_t w2 = 0, w1 = base;                        // steps k - 2 and k - 1 (step 0 is `base` itself: lanes 0, 1 final)
    base += dpp_row0<ROR + 15>(sha_s0(w1)) & lane15;   // lane 15 reads lane 0
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const uint32_t t = base + dpp_row0<SHR + 7>(w2);
        const uint32_t w0 = t + dpp_row0<SHR + 2>(sha_s1(w1));
        w2 = w1;
        w1 = w0;
    }
    return w1;
}
// The schedule of this row's block, published chunk by chunk like sha256_schedule_to_lds: `w` is lane i's message word i (rows of a
// wave: up to four blocks), kw / flag this row's (flag == nullptr: no counter, the caller synchronises); rows with active == false
// compute along and publish nothing.
__device__ __forceinline__ void sha256_schedule_rows_to_lds(uint32_t w, uint32_t* __restrict__ kw, volatile uint32_t* flag, uint32_t flag_base,
                                                         uint32_t first_chunk, bool active) {
    const uint32_t i = threadIdx.x & 15;
    if (first_chunk == 0 && active) {
        kw[i] = SHA256_K[i] + w;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the data before the counter
        if (flag && i == 0) *flag = flag_base + 1;
    }
#pragma unroll 1
    for (int base = 16; base < 64; base += 16) {
        w = sha256_schedule_chunk_rows(w);
        if (active) {
            kw[base + i] = SHA256_K[base + i] + w;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (flag && i == 0) *flag = flag_base + 1 + base / 16;
        }
    }
}

// ---- state rounds on SIX lanes ------------------------------------------------------------------------------------
// A lone wave pays for INSTRUCTIONS (4-6 cycles each whatever the dependencies), and a round above is 14 of them: six
// rotates, two xor3, Ch, Maj, four additions.  Its two halves -- Sigma1(e) + Ch(e, f, g) + h + kw and Sigma0(a) + Maj(a, b, c) --
// have the same shape, and the three rotates of a Sigma are one instruction with a per-lane amount.  So the state lives on
// six lanes of every row of 16: lanes 0, 1, 2 hold (e, f, g, h) and rotate by 6, 11, 25; lanes 4, 5, 6 hold (a, b, c, d)
// and rotate by 2, 13, 22.  One round:
//     x  = rotr(R0, s)                                  one v_alignbit, the lane's own amount
//     p  = R0 ^ (R2 & m)                                m = 0 on the e lanes, ~0 on the a lanes
//     F  = p ? R1 : R2 (bitwise)                        Ch(e, f, g) on the e lanes; Ch(a ^ c, b, c) = Maj(a, b, c) on the a lanes
//     Sg = x ^ x' ^ x''                                 two v_xor with a quad permutation (DPP): every lane of a triple gets the Sigma
//     T  = Sg + F + q                                   q = h + kw on the e lanes (kept one round ahead), 0 on the a lanes
//     e' = T + d (d from the lane four up), a' = T + T (T of the lane four down): two bank-masked DPP additions
// = 9 instructions, ~41 cycles instead of ~74.  The DPP reads observe the two wait states the ISA asks for after a VALU write
// of the register they read (x: p and F in between; T: the e-lane addition and the next q in between).
struct ShaSplit {
    uint32_t s, m;
    __device__ __forceinline__ void init() {
        const uint32_t l = threadIdx.x & 7;
        const bool a_lane = (l & 4) != 0;
        const uint32_t k = l & 3;
        s = a_lane ? (k == 0 ? 2u : k == 1 ? 13u : 22u) : (k == 0 ? 6u : k == 1 ? 11u : 25u);
        m = a_lane ? 0xFFFFFFFFu : 0u;
    }
    // h[0..7] (the same on every lane) -> R0..R3 of this lane's half
    __device__ __forceinline__ void split(const uint32_t (&h)[8], uint32_t (&hs)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) hs[i] = m ? h[i] : h[4 + i];
    }
    __device__ __forceinline__ void join(const uint32_t (&hs)[4], uint32_t (&h)[8]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = __builtin_amdgcn_readlane(hs[i], 4);
            h[4 + i] = __builtin_amdgcn_readlane(hs[i], 0);
        }
    }
};
// One asm statement per 16 rounds (between separate statements the compiler puts an s_nop of its own).  R3 <- the new R0;
// afterwards the roles are (R3, R0, R1, R2).  KWN: kw of the NEXT round (q runs one round ahead).
#define ZK_SHA_SPLIT_RND_HEAD(R0, R1, R2, R3)                                                          \
    "v_alignbit_b32 %[x], %[" #R0 "], %[" #R0 "], %[s]\n\t"                                            \
    "v_bitop3_b32 %[p], %[" #R0 "], %[" #R2 "], %[m] bitop3:0x78\n\t"                                  \
    "v_bitop3_b32 %[f], %[p], %[" #R1 "], %[" #R2 "] bitop3:0xca\n\t"                                  \
    "v_xor_b32_dpp %[t], %[x], %[x] quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n\t"                \
    "v_xor_b32_dpp %[t], %[x], %[t] quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n\t"                \
    "v_add3_u32 %[f], %[t], %[f], %[q]\n\t"                                                            \
    "v_add_u32_dpp %[" #R3 "], %[" #R3 "], %[f] row_shl:4 row_mask:0xf bank_mask:0x1\n\t"
#define ZK_SHA_SPLIT_RND(R0, R1, R2, R3, KWN)                                                          \
    ZK_SHA_SPLIT_RND_HEAD(R0, R1, R2, R3)                                                              \
    "v_add_u32_dpp %[q], %[" #R2 "], %[" #KWN "] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x1\n\t"   \
    "v_add_u32_dpp %[" #R3 "], %[f], %[f] row_shr:4 row_mask:0xf bank_mask:0x2\n\t"
// the last round of a chunk has no next kw at hand: a wait state in q's place
#define ZK_SHA_SPLIT_RND_LAST(R0, R1, R2, R3)                                                          \
    ZK_SHA_SPLIT_RND_HEAD(R0, R1, R2, R3)                                                              \
    "s_nop 0\n\t"                                                                                      \
    "v_add_u32_dpp %[" #R3 "], %[f], %[f] row_shr:4 row_mask:0xf bank_mask:0x2\n\t"
// q of the chunk's first round: h + kw on the e lanes (bank 0 of every row); the a lanes keep their 0
#define ZK_SHA_SPLIT_16ROUNDS(V)                                                                       \
    asm volatile("s_nop 1\n\t"                                                                         \
                 "v_add_u32_dpp %[q], %[r3], %[k0] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x1\n\t" \
                 ZK_SHA_SPLIT_RND(r0, r1, r2, r3, k1) ZK_SHA_SPLIT_RND(r3, r0, r1, r2, k2)             \
                 ZK_SHA_SPLIT_RND(r2, r3, r0, r1, k3) ZK_SHA_SPLIT_RND(r1, r2, r3, r0, k4)             \
                 ZK_SHA_SPLIT_RND(r0, r1, r2, r3, k5) ZK_SHA_SPLIT_RND(r3, r0, r1, r2, k6)             \
                 ZK_SHA_SPLIT_RND(r2, r3, r0, r1, k7) ZK_SHA_SPLIT_RND(r1, r2, r3, r0, k8)             \
                 ZK_SHA_SPLIT_RND(r0, r1, r2, r3, k9) ZK_SHA_SPLIT_RND(r3, r0, r1, r2, k10)            \
                 ZK_SHA_SPLIT_RND(r2, r3, r0, r1, k11) ZK_SHA_SPLIT_RND(r1, r2, r3, r0, k12)           \
                 ZK_SHA_SPLIT_RND(r0, r1, r2, r3, k13) ZK_SHA_SPLIT_RND(r3, r0, r1, r2, k14)           \
                 ZK_SHA_SPLIT_RND(r2, r3, r0, r1, k15) ZK_SHA_SPLIT_RND_LAST(r1, r2, r3, r0)           \
                 : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [q] "+v"(q), [x] "=&v"(x_), [p] "=&v"(p_),  \
                   [f] "=&v"(f_), [t] "=&v"(t_)                                                        \
                 : [s] "v"(sp.s), [m] "v"(sp.m), [k0] "v"(V[0]), [k1] "v"(V[1]), [k2] "v"(V[2]), [k3] "v"(V[3]), [k4] "v"(V[4]),   \
                   [k5] "v"(V[5]), [k6] "v"(V[6]), [k7] "v"(V[7]), [k8] "v"(V[8]), [k9] "v"(V[9]), [k10] "v"(V[10]),               \
                   [k11] "v"(V[11]), [k12] "v"(V[12]), [k13] "v"(V[13]), [k14] "v"(V[14]), [k15] "v"(V[15]));
// state rounds of one block on the split state hs (sha256_rounds_block's counterpart): every lane of the wave runs it.
// The kw chunks are read one chunk ahead of the rounds that use them (two register sets, no copies).
__device__ __forceinline__ void sha256_rounds_block_split(const ShaSplit& sp, uint32_t (&hs)[4], const uint32_t* __restrict__ kw) {
    uint32_t r0 = hs[0], r1 = hs[1], r2 = hs[2], r3 = hs[3];
    uint32_t q = 0, x_, p_, f_, t_;
    uint32_t va[16], vb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) va[j] = kw[j];
#pragma unroll 1
    for (int half = 0; half < 64; half += 32) {
#pragma unroll
        for (int j = 0; j < 16; ++j) vb[j] = kw[half + 16 + j];
        ZK_SHA_SPLIT_16ROUNDS(va)
#pragma unroll
        for (int j = 0; j < 16; ++j) va[j] = kw[((half + 32) & 63) + j];   // the second time round: chunk 0 again, unused
        ZK_SHA_SPLIT_16ROUNDS(vb)
    }
    hs[0] += r0; hs[1] += r1; hs[2] += r2; hs[3] += r3;
}
// sha256_compress_kw's counterpart.  With `blk`: rounds 0..15 from the message words while the schedule wave works (a chunk of
// it takes 0.2 us, sixteen rounds 0.3), chunk 1 when its counter says so, chunks 2 and 3 together (they are ready by then: one
// wait, one load latency).  Without: the whole schedule was published before the hash came here -- one wait, then the plain rounds.
__device__ __forceinline__ void sha256_compress_kw_split(const ShaSplit& sp, uint32_t (&hs)[4], const uint32_t* blk /* 16 words or nullptr */,
                                                      const uint32_t* __restrict__ kw, volatile uint32_t* flag, uint32_t flag_base) {
    if (!blk) {
        sha256_wait_flag(flag, flag_base + 4);
        sha256_rounds_block_split(sp, hs, kw);
        return;
    }
    uint32_t r0 = hs[0], r1 = hs[1], r2 = hs[2], r3 = hs[3];
    uint32_t q = 0, x_, p_, f_, t_;
    uint32_t va[16], vb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) va[j] = SHA256_K[j] + blk[j];
    ZK_SHA_SPLIT_16ROUNDS(va)
    sha256_wait_flag(flag, flag_base + 2);
#pragma unroll
    for (int j = 0; j < 16; ++j) va[j] = kw[16 + j];
    ZK_SHA_SPLIT_16ROUNDS(va)
    sha256_wait_flag(flag, flag_base + 4);
#pragma unroll
    for (int j = 0; j < 16; ++j) { va[j] = kw[32 + j]; vb[j] = kw[48 + j]; }
    ZK_SHA_SPLIT_16ROUNDS(va)
    ZK_SHA_SPLIT_16ROUNDS(vb)
    hs[0] += r0; hs[1] += r1; hs[2] += r2; hs[3] += r3;
}
// A whole padded message whose schedules other waves publish (block 0 chunk by chunk from its second chunk on, the later blocks
// complete when the hash reaches them): the closing kernels' hash wave.  h: in the initial value / out the digest, on every lane.
__device__ __forceinline__ void sha256_message_split(uint32_t (&h)[8], const uint32_t* msg, const uint32_t* kw, volatile uint32_t* kw_ready,
                                                  uint32_t n_blocks) {
    ShaSplit sp;
    sp.init();
    uint32_t hs[4];
    sp.split(h, hs);
    sha256_compress_kw_split(sp, hs, msg, kw, &kw_ready[0], 0u);
    for (uint32_t b = 1; b < n_blocks; ++b) {
        sha256_wait_flag(&kw_ready[b], 4u);
        sha256_rounds_block_split(sp, hs, kw + 64 * b);
    }
    sp.join(hs, h);
}

// One compression by a whole wave (every lane holds the same h and blk): the schedule on the sixteen lanes of row 0, the state rounds on
// six -- ~1.9 us instead of the ~3.2 us of sha256_compress on one lane.  lds: 64 words of scratch nobody else touches.
__device__ __forceinline__ void sha256_compress_wave(uint32_t (&h)[8], const uint32_t (&blk)[16], uint32_t* lds) {
    const uint32_t i = threadIdx.x & 15;
    uint32_t w = blk[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) w = i == (uint32_t)k ? blk[k] : w;
    sha256_schedule_rows_to_lds(w, lds, nullptr, 0u, 0u, (threadIdx.x & 63) < 16);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // lanes read what other lanes of the wave wrote
    ShaSplit sp;
    sp.init();
    uint32_t hs[4];
    sp.split(h, hs);
    sha256_rounds_block_split(sp, hs, lds);
    sp.join(hs, h);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // (the scratch is rewritten by the next call)
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
    // the same by a whole wave (sha256_compress_wave; every lane of the wave calls with the same state; lds: 64 words of scratch)
    __device__ __forceinline__ Fr challenge_fr_wave(uint32_t* lds) {
        const uint64_t bits = len * 8;
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
        sha256_compress_wave(h, buf, lds);
        uint32_t d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = h[i];
        init();
        commit_words8(d);                                        // 8 words pending: no compression
        Fr v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v.l[i] = d[7 - i];
        v.reduce_once();
        v.reduce_once();
        return fr_to_mont_outlined(v);
    }
};

}  // namespace zk

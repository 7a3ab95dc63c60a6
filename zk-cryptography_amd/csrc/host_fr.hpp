// host_fr.hpp -- host-side helpers of the composed sumcheck provers (product code, not the test oracle):
//   * BLS12-381 Fr Montgomery arithmetic, used once per prove to build the small interpolation matrices
//     (evaluations at x = 0..d  ->  coefficients) that the device transcript kernel applies each round;
//   * SHA-256, used by MultiComposedSumcheckProver::prove's initial absorption of every table's bytes
//     (multi_composed_sumcheck.rs:51-53): O(N) hashing is sequential by construction, so the GPU converts
//     the tables to canonical big-endian bytes in parallel and the host hashes the stream.
#pragma once
#include <stdint.h>
#include <string.h>
#include <cpuid.h>
#include <immintrin.h>

#include <vector>

namespace zkhost {

typedef unsigned __int128 u128;
struct Fr { uint64_t l[4]; };
static const uint64_t FR_P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const uint64_t FR_R1[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
static const uint64_t FR_R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
static const uint64_t FR_INV = 0xfffffffeffffffffULL;

inline bool fr_geq_p(const uint64_t* t) {
    for (int i = 3; i >= 0; --i) { if (t[i] > FR_P[i]) return true; if (t[i] < FR_P[i]) return false; }
    return true;
}
inline void fr_sub_p(uint64_t* t) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)t[i] - FR_P[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline Fr fr_zero() { Fr z; memset(z.l, 0, 32); return z; }
inline Fr fr_one() { Fr o; memcpy(o.l, FR_R1, 32); return o; }
inline Fr fr_add(const Fr& a, const Fr& b) {
    Fr r; uint64_t c = 0;
    for (int i = 0; i < 4; ++i) { u128 s = (u128)a.l[i] + b.l[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    if (c || fr_geq_p(r.l)) fr_sub_p(r.l);
    return r;
}
inline Fr fr_sub(const Fr& a, const Fr& b) {
    Fr r; uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) { uint64_t c = 0; for (int i = 0; i < 4; ++i) { u128 s = (u128)r.l[i] + FR_P[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
    return r;
}
inline Fr fr_mul(const Fr& a, const Fr& b) {
    uint64_t t[6] = {0};
    for (int i = 0; i < 4; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < 4; ++j) { u128 s = (u128)a.l[j] * b.l[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        u128 s = (u128)t[4] + c; t[4] = (uint64_t)s; t[5] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * FR_INV;
        s = (u128)m * FR_P[0] + t[0]; c = (uint64_t)(s >> 64);
        for (int j = 1; j < 4; ++j) { s = (u128)m * FR_P[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        s = (u128)t[4] + c; t[3] = (uint64_t)s; t[4] = t[5] + (uint64_t)(s >> 64);
    }
    if (t[4] || fr_geq_p(t)) fr_sub_p(t);
    Fr r; memcpy(r.l, t, 32); return r;
}
// out of Montgomery form: four reduction steps on (a, 0) -- half the products of fr_mul(a, 1); canonical (< r)
inline Fr fr_from_mont(const Fr& a) {
    uint64_t t[5] = {a.l[0], a.l[1], a.l[2], a.l[3], 0};
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[0] * FR_INV;
        u128 s = (u128)m * FR_P[0] + t[0];
        uint64_t c = (uint64_t)(s >> 64);
        for (int j = 1; j < 4; ++j) { s = (u128)m * FR_P[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        t[3] = c;
    }
    if (fr_geq_p(t)) fr_sub_p(t);
    Fr r; memcpy(r.l, t, 32); return r;
}
// the canonical integer of a Montgomery-form element as 32 big-endian bytes (Fr::into_bigint().to_bytes_be())
inline void fr_mont_to_be(const uint64_t* mont, uint8_t* be) {
    Fr v; memcpy(v.l, mont, 32);
    const Fr c = fr_from_mont(v);
    for (int i = 0; i < 4; ++i) { const uint64_t w = __builtin_bswap64(c.l[3 - i]); memcpy(be + 8 * i, &w, 8); }
}
inline Fr fr_from_u64(uint64_t v) { Fr c = fr_zero(); c.l[0] = v; Fr r2; memcpy(r2.l, FR_R2, 32); return fr_mul(c, r2); }
inline Fr fr_inv(const Fr& a) {   // Fermat
    uint64_t e[4]; memcpy(e, FR_P, 32); e[0] -= 2;
    Fr acc = fr_one();
    for (int i = 255; i >= 0; --i) { acc = fr_mul(acc, acc); if ((e[i / 64] >> (i % 64)) & 1) acc = fr_mul(acc, a); }
    return acc;
}

// M[k][i] = coefficient of x^k in the Lagrange basis polynomial L_i over the nodes 0..d  (row-major, (d+1)^2 entries)
inline std::vector<Fr> interpolation_matrix(int d) {
    const int n = d + 1;
    std::vector<Fr> m((size_t)n * n, fr_zero());
    for (int i = 0; i < n; ++i) {
        std::vector<Fr> poly(1, fr_one());
        Fr denom = fr_one();
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            std::vector<Fr> nxt(poly.size() + 1, fr_zero());
            Fr xj = fr_from_u64((uint64_t)j);
            for (size_t k = 0; k < poly.size(); ++k) {
                nxt[k] = fr_sub(nxt[k], fr_mul(poly[k], xj));
                nxt[k + 1] = fr_add(nxt[k + 1], poly[k]);
            }
            poly.swap(nxt);
            denom = fr_mul(denom, fr_sub(fr_from_u64((uint64_t)i), xj));
        }
        Fr dinv = fr_inv(denom);
        for (int k = 0; k < n; ++k) m[(size_t)k * n + i] = fr_mul(poly[k], dinv);
    }
    return m;
}

// ---- SHA-256 (streaming) ---------------------------------------------------------------------------------
// MultiComposedSumcheckProver::prove absorbs the bytes of every table before its first round (multi_composed_sumcheck.rs:51-53):
// one hash chain over K * N * 32 bytes, serial by construction, so it runs on the host -- with the SHA extensions where the CPU
// has them (~2 GB/s per core instead of ~0.4 GB/s for the portable rounds; the EPYC hosts of MI355X nodes do).
alignas(16) static const uint32_t SHA256_K_HOST[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
inline bool cpu_has_sha_ext() {
    static const bool has = [] {
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (!__get_cpuid(1, &a, &b, &c, &d) || !(c & (1u << 9)) || !(c & (1u << 19))) return false;   // SSSE3, SSE4.1
        if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
        return (b & (1u << 29)) != 0;                                                                  // SHA
    }();
    return has;
}
// n_blocks whole 64-byte blocks into the state h[8] = (a..h).  W_i (four words) = msg2(msg1(W_{i-4}, W_{i-3}) + (W_{i-2}:W_{i-1} >> 32), W_{i-1}).
__attribute__((target("sha,sse4.1,ssse3"))) inline void sha256_blocks_sha_ext(uint32_t h[8], const uint8_t* data, size_t n_blocks) {
    const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
    __m128i t = _mm_loadu_si128((const __m128i*)&h[0]);      // d c b a (high .. low)
    __m128i s1 = _mm_loadu_si128((const __m128i*)&h[4]);     // h g f e
    t = _mm_shuffle_epi32(t, 0xB1);                          // c d a b
    s1 = _mm_shuffle_epi32(s1, 0x1B);                        // e f g h
    __m128i s0 = _mm_alignr_epi8(t, s1, 8);                  // a b e f
    s1 = _mm_blend_epi16(s1, t, 0xF0);                       // c d g h
    for (; n_blocks; --n_blocks, data += 64) {
        const __m128i save0 = s0, save1 = s1;
        __m128i m[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __m128i w;
            if (i < 4) {
                w = m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(data + 16 * i)), bswap);
            } else {
                __m128i x = _mm_sha256msg1_epu32(m[(i - 4) & 3], m[(i - 3) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(m[(i - 1) & 3], m[(i - 2) & 3], 4));
                w = m[i & 3] = _mm_sha256msg2_epu32(x, m[(i - 1) & 3]);
            }
            __m128i kw = _mm_add_epi32(w, _mm_load_si128((const __m128i*)&SHA256_K_HOST[4 * i]));
            s1 = _mm_sha256rnds2_epu32(s1, s0, kw);
            kw = _mm_shuffle_epi32(kw, 0x0E);
            s0 = _mm_sha256rnds2_epu32(s0, s1, kw);
        }
        s0 = _mm_add_epi32(s0, save0);
        s1 = _mm_add_epi32(s1, save1);
    }
    t = _mm_shuffle_epi32(s0, 0x1B);                         // f e b a
    s1 = _mm_shuffle_epi32(s1, 0xB1);                        // d c h g
    s0 = _mm_blend_epi16(t, s1, 0xF0);                       // d c b a
    s1 = _mm_alignr_epi8(s1, t, 8);                          // h g f e
    _mm_storeu_si128((__m128i*)&h[0], s0);
    _mm_storeu_si128((__m128i*)&h[4], s1);
}

struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    uint64_t len;
    Sha256() { reset(); }
    void reset() {
        static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        memcpy(h, iv, 32); len = 0;
    }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void compress(const uint8_t* b) {
        const uint32_t* K = SHA256_K_HOST;
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t* d, size_t n) {
        size_t fill = (size_t)(len % 64);
        len += n;
        if (fill) {
            size_t take = 64 - fill; if (take > n) take = n;
            memcpy(buf + fill, d, take); d += take; n -= take; fill += take;
            if (fill == 64) compress(buf); else return;
        }
        if (n >= 64 && cpu_has_sha_ext()) {
            const size_t blocks = n / 64;
            sha256_blocks_sha_ext(h, d, blocks);
            d += 64 * blocks; n -= 64 * blocks;
        }
        while (n >= 64) { compress(d); d += 64; n -= 64; }
        if (n) memcpy(buf, d, n);
    }
    void finish(uint8_t out[32]) {   // digest of everything absorbed; the object is consumed (reset() to reuse)
        const uint64_t bits = len * 8;
        uint8_t pad[72] = {0x80};
        const size_t fill = (size_t)(len % 64), padlen = (fill < 56 ? 56 : 120) - fill;
        update(pad, padlen);
        uint8_t lb[8];
        for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(lb, 8);
        for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16); out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i]; }
    }
};

// FiatShamirTranscript on the host (transcripts/fiat-shamir/src/fiat_shamir.rs:10-40): the GKR outer transcript
struct Transcript {
    Sha256 hasher;
    void commit(const uint8_t* d, size_t n) { hasher.update(d, n); }
    void challenge(uint8_t out[32]) {            // finalize_reset + update(digest)  :21-25
        hasher.finish(out);
        hasher.reset();
        hasher.update(out, 32);
    }
    Fr challenge_fr() {                           // from_be_bytes_mod_order(digest)  :27-29, Montgomery form
        uint8_t d[32];
        challenge(d);
        uint64_t t[4];
        for (int i = 0; i < 4; ++i) {
            uint64_t w = 0;
            for (int j = 0; j < 8; ++j) w = (w << 8) | d[8 * (3 - i) + j];
            t[i] = w;
        }
        while (fr_geq_p(t)) fr_sub_p(t);          // 2^256 < 3 p: at most two subtractions
        Fr c, r2;
        memcpy(c.l, t, 32);
        memcpy(r2.l, FR_R2, 32);
        return fr_mul(c, r2);
    }
};

}  // namespace zkhost

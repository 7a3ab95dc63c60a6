// wide_acc.hpp -- unreduced accumulation of Fr products (gfx950): sum_i a_i * b_i is kept as a 17-limb integer and reduced
// once, so a term costs the 64 mads + carry adds of the schoolbook product and none of the Montgomery reduction.
// Used by the k-variable fold (multifold_kernels.hpp) and by the K = 2 round sums of large composed claims
// (composed_kernels.hpp).
#pragma once
#include "fp.hpp"

namespace zk {

// Unreduced accumulator: column c collects every limb product w[i]*t[j] with i + j = c.
struct WideAcc {
    uint64_t lo[15];
    uint32_t hi[15];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int c = 0; c < 15; ++c) { lo[c] = 0; hi[c] = 0; }
    }
    __device__ __forceinline__ void mac(const Fr& w, const Fr& t) {
#pragma unroll
        for (int c = 0; c < 15; ++c) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = c - i;
                if (j >= 0 && j < 8) mac96(lo[c], hi[c], w.l[i], t.l[j]);
            }
        }
    }
};

// 9-word Montgomery reduction of a 17-limb integer x (x[17] = 0 on entry): returns x * 2^-288 mod r, canonical
__device__ __forceinline__ Fr wide_redc(uint32_t (&x)[18]) {
    // word-serial REDC, 9 words: after step i, limb i is zero
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t m = FrParams::mul_inv(x[i]);
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t s = (uint64_t)m * FrParams::p(j) + x[i + j] + c;
            x[i + j] = (uint32_t)s;
            c = s >> 32;
        }
#pragma unroll
        for (int j = i + 8; j < 18; ++j) {
            uint64_t s = (uint64_t)x[j] + c;
            x[j] = (uint32_t)s;
            c = s >> 32;
        }
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = x[9 + i];
    // x < 2^520  =>  x / 2^288 + r < 2r: one conditional subtraction (x[17] is zero)
    r.reduce_once();
    return r;
}

// x = sum_c col[c] * 2^(32c) as 17 limbs, then 9-word Montgomery reduction: returns x * 2^-288 mod r, canonical
__device__ __forceinline__ Fr wide_reduce(const uint64_t (&lo)[15], const uint32_t (&hi)[15]) {
    uint32_t x[18];
    uint64_t carry = 0;     // running value above the current limb (< 2^64 + small)
    uint32_t carry_hi = 0;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        // add column c (96 bits) at limb c to the running carry (carry: 64 bit + carry_hi: 32 bit)
        uint64_t s = carry + lo[c];
        uint32_t ov = s < carry ? 1u : 0u;
        x[c] = (uint32_t)s;
        carry = (s >> 32) | ((uint64_t)(carry_hi + hi[c] + ov) << 32);
        carry_hi = 0;   // (carry_hi + hi[c] + ov) < 2^32: hi[c] counts at most 2^11 carries
    }
    x[15] = (uint32_t)carry;
    x[16] = (uint32_t)(carry >> 32);
    x[17] = 0;
    return wide_redc(x);
}

// Montgomery form of 2^32: wide_reduce divides by 2^288, so either one operand of every product carries this factor
// (the fold weights) or the reduced sum is multiplied by it once.
__device__ __forceinline__ Fr fr_mont_2_32() {
    constexpr uint32_t c[8] = {0xcaaf6b13u, 0x355094eau, 0x69a568efu, 0xf6b10cb3u, 0x40cc3869u, 0xe2c926a6u, 0xed269aadu, 0x736a6d3bu};
    Fr w;
#pragma unroll
    for (int i = 0; i < 8; ++i) w.l[i] = c[i];
    return w;
}

}  // namespace zk

// fqu.hpp -- BLS12-381 Fq in an UNSATURATED representation for the MSM inner loops (gfx950).
//
// 14 limbs of 28 bits, Montgomery radix R = 2^392.  On gfx950 v_mad_u64_u32 costs the same issue slot as a
// plain multiply or a carry add (profiles/r01/ubench_alu_gfx950.txt), so what a saturated product pays for its
// 288 carry adds is pure overhead: with 28-bit limbs a whole column of limb products (<= 28 of them, < 2^58
// each) fits one 64-bit accumulator and the product is 406 mads and nothing else.  R is 11 bits above p, which
// also makes lazy reduction free: values up to ~64 p are legal multiplier inputs and the product is always < 2p,
// so additions are 14 independent adds and subtractions add a precomputed multiple of p instead of branching.
//
// Conventions
//   value  : the integer sum l[i] 2^(28 i), congruent to x * 2^392 (mod p); NOT necessarily < p.
//   limbs  : "weak" = limbs 0..12 < 2^28 + 2^4 (what sub/weak_norm return), "normalised" = limbs 0..12 < 2^28
//            (what mul returns).  Multiplier inputs may carry limbs up to 2^29.5.
// Points enter from / leave to the arkworks layout (12 x 32-bit saturated limbs, R = 2^384) through
// from_ark / to_ark; parity with the reference is checked on the converted, canonical values.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fp.hpp"
#include "fqu_consts.hpp"

namespace zk {

struct FqU {
    static constexpr int N = 14;
    static constexpr uint32_t MASK = (1u << 28) - 1;
    uint32_t l[N];

    __device__ __forceinline__ static FqU zero() {
        FqU r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = 0;
        return r;
    }
    __device__ __forceinline__ static FqU one() {
        FqU r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = FquConsts::ONE[i];
        return r;
    }
    __device__ __forceinline__ bool all_zero() const {   // exact zero limbs (the stored identity marker)
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) x |= l[i];
        return x == 0;
    }
};

// limbs 0..12 back under 2^28 + 2^4, all carries taken in parallel
__device__ __forceinline__ FqU fqu_weak_norm(const FqU& x) {
    FqU r;
    r.l[0] = x.l[0] & FqU::MASK;
#pragma unroll
    for (int i = 1; i < FqU::N - 1; ++i) r.l[i] = (x.l[i] & FqU::MASK) + (x.l[i - 1] >> 28);
    r.l[FqU::N - 1] = x.l[FqU::N - 1] + (x.l[FqU::N - 2] >> 28);
    return r;
}
// exact carry propagation: limbs 0..12 < 2^28
__device__ __forceinline__ FqU fqu_strong_norm(const FqU& x) {
    FqU r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < FqU::N - 1; ++i) {
        uint32_t v = x.l[i] + c;
        r.l[i] = v & FqU::MASK;
        c = v >> 28;
    }
    r.l[FqU::N - 1] = x.l[FqU::N - 1] + c;
    return r;
}

__device__ __forceinline__ FqU fqu_add(const FqU& a, const FqU& b) {   // lazy: limbs add up, no carries
    FqU r;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
__device__ __forceinline__ FqU fqu_dbl(const FqU& a) { return fqu_add(a, a); }

// a - b + K p, K in {4, 8, 16}: the redundant limbs of K p dominate any weakly normalised subtrahend, so no limb goes
// negative; requires value(b) < (K - 1) p.  Result weakly normalised, value < value(a) + K p.
template <int K>
__device__ __forceinline__ FqU fqu_sub(const FqU& a, const FqU& b) {
    FqU r;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) {
        const uint32_t kp = (K == 4) ? FquConsts::P4_RED[i] : (K == 8) ? FquConsts::P8_RED[i] : FquConsts::P16_RED[i];
        r.l[i] = a.l[i] + kp - b.l[i];
    }
    return fqu_weak_norm(r);
}
__device__ __forceinline__ FqU fqu_neg4(const FqU& b) { return fqu_sub<4>(FqU::zero(), b); }   // value(b) < 3p

// Montgomery product a b 2^-392: one 64-bit accumulator per column, 406 mads.  Inputs: limbs < 2^29.5, values
// < 64 p.  Output normalised, value < 2p.
__device__ __forceinline__ FqU fqu_mul_inline(const FqU& a, const FqU& b) {
    constexpr int N = FqU::N;
    uint32_t m[N];
    FqU r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * FquConsts::P[k - i];
        m[k] = ((uint32_t)acc * FquConsts::INV) & FqU::MASK;
        acc += (uint64_t)m[k] * FquConsts::P[0];
        acc >>= 28;
    }
#pragma unroll
    for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
        for (int i = k - N + 1; i < N; ++i) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)m[i] * FquConsts::P[k - i];
        }
        r.l[k - N] = (uint32_t)acc & FqU::MASK;
        acc >>= 28;
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
}
#ifndef ZK_FQU_INLINE
__device__ __noinline__ FqU fqu_mul(FqU a, FqU b) { return fqu_mul_inline(a, b); }
#else
__device__ __forceinline__ FqU fqu_mul(const FqU& a, const FqU& b) { return fqu_mul_inline(a, b); }
#endif
__device__ __forceinline__ FqU fqu_sqr(const FqU& a) { return fqu_mul(a, a); }

// exact test value(x) == t * p for a small t (slow path of fqu_is_zero_mod_p)
__device__ __noinline__ bool fqu_equals_multiple(FqU x, uint32_t t) {
    FqU s = fqu_strong_norm(x);
    uint64_t c = 0;
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < FqU::N - 1; ++i) {
        c += (uint64_t)t * FquConsts::P[i];
        diff |= ((uint32_t)c & FqU::MASK) ^ s.l[i];
        c >>= 28;
    }
    c += (uint64_t)t * FquConsts::P[FqU::N - 1];
    diff |= (uint32_t)c ^ s.l[FqU::N - 1];
    return diff == 0 && (c >> 32) == 0;
}
// value(x) congruent to 0 mod p?  value(x) < 32 p.  If x = t p then its low 28 bits times p^-1 give t back, so
// anything else is rejected by one multiply (false positives, probability 2^-23, fall through to the exact test).
__device__ __forceinline__ bool fqu_is_zero_mod_p(const FqU& x) {
    const uint32_t t = ((x.l[0] & FqU::MASK) * FquConsts::PINV) & FqU::MASK;
    if (t >= 32) return false;
    return fqu_equals_multiple(x, t);
}

// arkworks form (12 x 32 saturated, x 2^384, canonical) -> internal
__device__ __forceinline__ FqU fqu_from_ark(const Fq& a) {
    FqU v;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) {
        const int bit = 28 * i, w = bit >> 5, sh = bit & 31;
        uint64_t two = a.l[w];
        if (w + 1 < 12) two |= (uint64_t)a.l[w + 1] << 32;
        v.l[i] = (uint32_t)(two >> sh) & FqU::MASK;
    }
    FqU c;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) c.l[i] = FquConsts::C_IN[i];
    return fqu_mul(v, c);
}
// internal (value < 32 p) -> arkworks canonical limbs
__device__ __forceinline__ Fq fqu_to_ark(const FqU& x) {
    FqU c;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) c.l[i] = FquConsts::C_OUT[i];
    FqU y = fqu_strong_norm(fqu_mul(x, c));     // value < 2p, limbs exact
    // conditional subtraction of p
    FqU d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) {
        uint32_t v = y.l[i] - FquConsts::P[i] - borrow;
        borrow = (i < FqU::N - 1) ? (v >> 31) : (v >> 31);
        d.l[i] = (i < FqU::N - 1) ? (v & FqU::MASK) : v;
    }
    if (!borrow) y = d;
    Fq r;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int bit = 32 * j, li = bit / 28, off = bit % 28;
        uint64_t v = (uint64_t)y.l[li] >> off;
        if (li + 1 < FqU::N) v |= (uint64_t)y.l[li + 1] << (28 - off);
        if (li + 2 < FqU::N) v |= (uint64_t)y.l[li + 2] << (56 - off);
        r.l[j] = (uint32_t)v;
    }
    return r;
}

// memory layout of an internal element: 16 x u32 (14 limbs + 2 pad) = 64 bytes
__device__ __forceinline__ FqU load_fqu(const uint32_t* __restrict__ p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    FqU r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    r.l[12] = d.x; r.l[13] = d.y;
    return r;
}
__device__ __forceinline__ void store_fqu(uint32_t* __restrict__ p, const FqU& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
    q[3] = make_uint4(v.l[12], v.l[13], 0u, 0u);
}

}  // namespace zk

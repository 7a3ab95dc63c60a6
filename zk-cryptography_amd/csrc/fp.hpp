// fp.hpp -- Montgomery prime-field arithmetic for gfx950 (CDNA4), device side.
//
// Replaces, on the GPU, the ark-ff `Fp<MontBackend<_, N>, N>` operations the
// reference reaches from polynomial/src/multilinear/evaluation_form.rs:133
// (Fr mul/add/sub) and from `mul_bigint` in kzg/src/univariate_kzg.rs:53 /
// kzg/src/multilinear_kzg.rs:46 (Fq inside the G1 group law).
//
// Representation: N32 saturated 32-bit limbs, little-endian, Montgomery form with
// R = 2^(32*N32) -- bit-identical in memory to arkworks' [u64; N32/2].  All
// results are fully reduced (canonical Montgomery residues), so equality with the
// reference is equality of limbs.
//
// The multiplier is product-scanning (FIPS) Montgomery: one 96-bit column
// accumulator fed by v_mad_u64_u32 (32x32+64 -> 64, carry-out in VCC) followed by
// a v_addc_co_u32 into the top word.  2*N32^2 mads per product; the modulus limbs
// ride in SGPRs (constant bus) so they cost no VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zk {

// acc(96 bit: lo64, hi32) += a * b
__device__ __forceinline__ void mac96(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "v"(b)
        : "vcc");
}
// same with a wave-uniform multiplier held in an SGPR (modulus limb)
__device__ __forceinline__ void mac96_s(uint64_t& lo, uint32_t& hi, uint32_t s_const, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "s"(s_const), "v"(b)
        : "vcc");
}


// ---- carry chains of 8 limbs (Fr), one asm statement each so that VCC links the whole chain ------------------------
// (The same loops written with 64-bit C arithmetic compile to ~100 instructions per modular addition -- v_lshl_add_u64 plus
// moves that rebuild the carries -- instead of these 8 + 8 + 8.)
// r += b   (no carry out of the top limb by construction: the moduli leave spare bits)
__device__ __forceinline__ void add_chain8(uint32_t (&r)[8], const uint32_t (&b)[8]) {
    asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %15, vcc"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
        : "vcc");
}
// r = (r >= p) ? r - p : r      (p in VGPRs: a carry-in from VCC already takes the instruction's one scalar operand)
__device__ __forceinline__ void cond_sub_chain8(uint32_t (&r)[8], const uint32_t (&p)[8]) {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_sub_co_u32 %8, vcc, %0, %16\n\t"
        "v_subb_co_u32 %9, vcc, %1, %17, vcc\n\t"
        "v_subb_co_u32 %10, vcc, %2, %18, vcc\n\t"
        "v_subb_co_u32 %11, vcc, %3, %19, vcc\n\t"
        "v_subb_co_u32 %12, vcc, %4, %20, vcc\n\t"
        "v_subb_co_u32 %13, vcc, %5, %21, vcc\n\t"
        "v_subb_co_u32 %14, vcc, %6, %22, vcc\n\t"
        "v_subb_co_u32 %15, vcc, %7, %23, vcc\n\t"
        "v_cndmask_b32 %0, %8, %0, vcc\n\t"
        "v_cndmask_b32 %1, %9, %1, vcc\n\t"
        "v_cndmask_b32 %2, %10, %2, vcc\n\t"
        "v_cndmask_b32 %3, %11, %3, vcc\n\t"
        "v_cndmask_b32 %4, %12, %4, vcc\n\t"
        "v_cndmask_b32 %5, %13, %5, vcc\n\t"
        "v_cndmask_b32 %6, %14, %6, vcc\n\t"
        "v_cndmask_b32 %7, %15, %7, vcc"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "vcc");
}
// r -= b, returns the borrow as a mask (all ones when r < b)
__device__ __forceinline__ uint32_t sub_chain8(uint32_t (&r)[8], const uint32_t (&b)[8]) {
    uint32_t mask;
    asm("v_sub_co_u32 %0, vcc, %0, %9\n\t"
        "v_subb_co_u32 %1, vcc, %1, %10, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %2, %11, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %3, %12, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %4, %13, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %5, %14, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %6, %15, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %7, %16, vcc\n\t"
        "v_cndmask_b32 %8, 0, -1, vcc"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "=&v"(mask)
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
        : "vcc");
    return mask;
}

// ---- the same for 12 limbs (Fq) ----
__device__ __forceinline__ void add_chain12(uint32_t (&r)[12], const uint32_t (&b)[12]) {
    asm("v_add_co_u32 %0, vcc, %0, %12\n\t"
        "v_addc_co_u32 %1, vcc, %1, %13, vcc\n\t"
        "v_addc_co_u32 %2, vcc, %2, %14, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %15, vcc\n\t"
        "v_addc_co_u32 %4, vcc, %4, %16, vcc\n\t"
        "v_addc_co_u32 %5, vcc, %5, %17, vcc\n\t"
        "v_addc_co_u32 %6, vcc, %6, %18, vcc\n\t"
        "v_addc_co_u32 %7, vcc, %7, %19, vcc\n\t"
        "v_addc_co_u32 %8, vcc, %8, %20, vcc\n\t"
        "v_addc_co_u32 %9, vcc, %9, %21, vcc\n\t"
        "v_addc_co_u32 %10, vcc, %10, %22, vcc\n\t"
        "v_addc_co_u32 %11, vcc, %11, %23, vcc"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11])
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11])
        : "vcc");
}
__device__ __forceinline__ uint32_t sub_chain12(uint32_t (&r)[12], const uint32_t (&b)[12]) {
    uint32_t mask;
    asm("v_sub_co_u32 %0, vcc, %0, %13\n\t"
        "v_subb_co_u32 %1, vcc, %1, %14, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %2, %15, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %3, %16, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %4, %17, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %5, %18, vcc\n\t"
        "v_subb_co_u32 %6, vcc, %6, %19, vcc\n\t"
        "v_subb_co_u32 %7, vcc, %7, %20, vcc\n\t"
        "v_subb_co_u32 %8, vcc, %8, %21, vcc\n\t"
        "v_subb_co_u32 %9, vcc, %9, %22, vcc\n\t"
        "v_subb_co_u32 %10, vcc, %10, %23, vcc\n\t"
        "v_subb_co_u32 %11, vcc, %11, %24, vcc\n\t"
        "v_cndmask_b32 %12, 0, -1, vcc"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "=&v"(mask)
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11])
        : "vcc");
    return mask;
}
// t = r - p in two statements of 6 limbs (36 operands do not fit one), the borrow handed over in a VGPR; returns the
// final borrow as a mask (all ones when r < p)
__device__ __forceinline__ uint32_t sub_to_chain12(uint32_t (&t)[12], const uint32_t (&r)[12], const uint32_t (&p)[12]) {
    uint32_t bor, mask;
    asm("v_sub_co_u32 %0, vcc, %7, %13\n\t"
        "v_subb_co_u32 %1, vcc, %8, %14, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %9, %15, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %10, %16, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %11, %17, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %12, %18, vcc\n\t"
        "v_cndmask_b32 %6, 0, 1, vcc"
        : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(bor)
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5])
        : "vcc");
    asm("v_cmp_ne_u32 vcc, 0, %7\n\t"
        "v_subb_co_u32 %0, vcc, %8, %14, vcc\n\t"
        "v_subb_co_u32 %1, vcc, %9, %15, vcc\n\t"
        "v_subb_co_u32 %2, vcc, %10, %16, vcc\n\t"
        "v_subb_co_u32 %3, vcc, %11, %17, vcc\n\t"
        "v_subb_co_u32 %4, vcc, %12, %18, vcc\n\t"
        "v_subb_co_u32 %5, vcc, %13, %19, vcc\n\t"
        "v_cndmask_b32 %6, 0, -1, vcc"
        : "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(mask)
        : "v"(bor), "v"(r[6]), "v"(r[7]), "v"(r[8]), "v"(r[9]), "v"(r[10]), "v"(r[11]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11])
        : "vcc");
    return mask;
}

}  // namespace zk
#include "fp_mul_gen.hpp"
namespace zk {

template <class P>
struct Fp {
    static constexpr int N = P::N32;
    uint32_t l[N];

    __device__ __forceinline__ static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = 0;
        return r;
    }
    __device__ __forceinline__ static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.l[i] = P::r1(i);
        return r;
    }
    __device__ __forceinline__ bool is_zero() const {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) x |= l[i];
        return x == 0;
    }
    __device__ __forceinline__ bool operator==(const Fp& o) const {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) x |= l[i] ^ o.l[i];
        return x == 0;
    }

    // r = a - p if a >= p (a < 2p)
    __device__ __forceinline__ void reduce_once() {
        if constexpr (N == 8) {
            uint32_t p[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = P::p(i);
            cond_sub_chain8(l, p);
            return;
        } else if constexpr (N == 12) {
            uint32_t p[12], t[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) p[i] = P::p(i);
            const uint32_t keep = sub_to_chain12(t, l, p);   // all ones: l < p, keep l
#pragma unroll
            for (int i = 0; i < 12; ++i) l[i] = keep ? l[i] : t[i];
            return;
        }
        uint32_t t[N];
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t d = (uint64_t)l[i] - P::p(i) - borrow;
            t[i] = (uint32_t)d;
            borrow = (d >> 32) & 1;
        }
        if (!borrow) {
#pragma unroll
            for (int i = 0; i < N; ++i) l[i] = t[i];
        }
    }

    __device__ __forceinline__ friend Fp operator+(const Fp& a, const Fp& b) {
        if constexpr (N == 8) {
            Fp r = a;
            add_chain8(r.l, b.l);
            r.reduce_once();
            return r;
        } else if constexpr (N == 12) {
            Fp r = a;
            add_chain12(r.l, b.l);
            r.reduce_once();
            return r;
        }
        Fp r;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            c += (uint64_t)a.l[i] + b.l[i];
            r.l[i] = (uint32_t)c;
            c >>= 32;
        }
        // P::SPARE_BITS >= 1: a + b < 2p < 2^(32N), no carry out
        r.reduce_once();
        return r;
    }
    __device__ __forceinline__ friend Fp operator-(const Fp& a, const Fp& b) {
        if constexpr (N == 8) {
            Fp r = a;
            const uint32_t mask = sub_chain8(r.l, b.l);
            uint32_t pm[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pm[i] = P::p(i) & mask;   // add p back when a < b
            add_chain8(r.l, pm);
            return r;
        } else if constexpr (N == 12) {
            Fp r = a;
            const uint32_t mask = sub_chain12(r.l, b.l);
            uint32_t pm[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) pm[i] = P::p(i) & mask;
            add_chain12(r.l, pm);
            return r;
        }
        Fp r;
        uint64_t borrow = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
            r.l[i] = (uint32_t)d;
            borrow = (d >> 32) & 1;
        }
        // add p back when a < b (mask instead of branch: lanes diverge on data)
        uint32_t mask = 0u - (uint32_t)borrow;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            c += (uint64_t)r.l[i] + (P::p(i) & mask);
            r.l[i] = (uint32_t)c;
            c >>= 32;
        }
        return r;
    }
    __device__ __forceinline__ Fp neg() const { return zero() - *this; }
    __device__ __forceinline__ Fp dbl() const { return *this + *this; }

    // Montgomery product a*b*R^-1 mod p, product scanning.
    __device__ __forceinline__ friend Fp operator*(const Fp& a, const Fp& b) {
        if constexpr (N == 8) {
            Fp r;
            mont_mul_unrolled_8<P>(r.l, a.l, b.l);
            r.reduce_once();
            return r;
        } else if constexpr (N == 12) {
            Fp r;
            mont_mul_unrolled_12<P>(r.l, a.l, b.l);
            r.reduce_once();
            return r;
        }
        uint32_t m[N];
        Fp r;
        uint64_t lo = 0;
        uint32_t hi = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
#pragma unroll
            for (int i = 0; i < k; ++i) {
                mac96(lo, hi, a.l[i], b.l[k - i]);
                mac96_s(lo, hi, P::p(k - i), m[i]);
            }
            mac96(lo, hi, a.l[k], b.l[0]);
            m[k] = P::mul_inv((uint32_t)lo);
            mac96_s(lo, hi, P::p(0), m[k]);
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
#pragma unroll
        for (int k = N; k < 2 * N - 1; ++k) {
#pragma unroll
            for (int i = k - N + 1; i < N; ++i) {
                mac96(lo, hi, a.l[i], b.l[k - i]);
                mac96_s(lo, hi, P::p(k - i), m[i]);
            }
            r.l[k - N] = (uint32_t)lo;
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
        r.l[N - 1] = (uint32_t)lo;
        // inputs < p and p < 2^(32N-1)  =>  result < 2p < 2^(32N): no overflow word
        r.reduce_once();
        return r;
    }
    __device__ __forceinline__ Fp sqr() const { return (*this) * (*this); }

    // Montgomery form -> canonical integer (into_bigint): multiply by 1
    __device__ __forceinline__ Fp from_mont() const {
        Fp o = zero();
        o.l[0] = 1;
        return (*this) * o;
    }
    // canonical integer (< p) -> Montgomery form: multiply by R^2
    __device__ __forceinline__ Fp to_mont() const {
        Fp r2;
#pragma unroll
        for (int i = 0; i < N; ++i) r2.l[i] = P::r2(i);
        return (*this) * r2;
    }
};

// ---- BLS12-381 scalar field Fr (255 bit), 8 x u32 -----------------------------------
struct FrParams {
    static constexpr int N32 = 8;
    static constexpr int SPARE_BITS = 1;
    __device__ __forceinline__ static constexpr uint32_t p(int i) {
        constexpr uint32_t v[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return v[i];
    }
    __device__ __forceinline__ static constexpr uint32_t r1(int i) {   // R mod r
        constexpr uint32_t v[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return v[i];
    }
    __device__ __forceinline__ static constexpr uint32_t r2(int i) {   // R^2 mod r
        constexpr uint32_t v[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return v[i];
    }
    // -r^-1 mod 2^32 = 0xffffffff  =>  x * inv = -x
    __device__ __forceinline__ static uint32_t mul_inv(uint32_t x) { return 0u - x; }
};

// ---- BLS12-381 base field Fq (381 bit), 12 x u32 --------------------------------------
struct FqParams {
    static constexpr int N32 = 12;
    static constexpr int SPARE_BITS = 3;
    __device__ __forceinline__ static constexpr uint32_t p(int i) {
        constexpr uint32_t v[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                    0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return v[i];
    }
    __device__ __forceinline__ static constexpr uint32_t r1(int i) {
        constexpr uint32_t v[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                    0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return v[i];
    }
    __device__ __forceinline__ static constexpr uint32_t r2(int i) {
        constexpr uint32_t v[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                    0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return v[i];
    }
    __device__ __forceinline__ static uint32_t mul_inv(uint32_t x) { return x * 0xfffcfffdu; }
};

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// ---- global memory access: one element = N32*4 contiguous bytes (arkworks layout) -------
__device__ __forceinline__ Fr load_fr(const uint64_t* __restrict__ base, size_t idx) {
    const uint4* p = reinterpret_cast<const uint4*>(base + 4 * idx);
    uint4 a = p[0], b = p[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
// once-read streaming load of a table entry (nt: no reuse to keep in the caches)
__device__ __forceinline__ Fr load_fr_nt(const uint64_t* __restrict__ base) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* p = reinterpret_cast<const u32x4*>(base);
    const u32x4 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1);
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ Fr load_fr_nt(const uint64_t* __restrict__ base, size_t idx) { return load_fr_nt(base + 4 * idx); }
__device__ __forceinline__ void store_fr(uint64_t* __restrict__ base, size_t idx, const Fr& v) {
    uint4* p = reinterpret_cast<uint4*>(base + 4 * idx);
    p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// ---- wavefront (64 lanes) + workgroup tree reduction with modular add ---------------------
__device__ __forceinline__ Fr shfl_down_fr(const Fr& v, int delta) {
    Fr r;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) r.l[i] = __shfl_down(v.l[i], delta, 64);
    return r;
}
__device__ __forceinline__ Fr shfl_fr(const Fr& v, int src_lane) {
    Fr r;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) r.l[i] = __shfl(v.l[i], src_lane, 64);
    return r;
}
// Cross-lane moves of a reduction as DPP / permlane VALU instructions, not ds_bpermute: a shuffle tree over a field element is 48
// LDS crossbar operations per value, and the block-sum pass (two values per 16 KiB per wave) kept every CU's LDS busy with them --
// fine_sums ran at 94-96 us where its loads alone take 81 (tools/ubench_rows.hip).
// row_shr:n (0x110 + n) within rows of 16 lanes, row_bcast:15 (0x142) / row_bcast:31 (0x143) across rows; lanes without a source, and
// rows masked out, contribute zero.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Fr dpp_fr(const Fr& v) {
    Fr r;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], CTRL, ROW_MASK, 0xf, true);
    return r;
}
__device__ __forceinline__ Fr readlane_fr(const Fr& v, int lane) {
    Fr r;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_readlane((int)v.l[i], lane);
    return r;
}
__device__ __forceinline__ Fr wave_reduce_fr(Fr v) {
    v = v + dpp_fr<0x111, 0xf>(v);
    v = v + dpp_fr<0x112, 0xf>(v);
    v = v + dpp_fr<0x114, 0xf>(v);
    v = v + dpp_fr<0x118, 0xf>(v);      // lane 15 of every row: the row's sum
    v = v + dpp_fr<0x142, 0xa>(v);      // rows 1 and 3 += lane 15 of the row before
    v = v + dpp_fr<0x143, 0xc>(v);      // rows 2 and 3 += lane 31
    return readlane_fr(v, 63);          // every lane holds the sum
}
// Two sums for six steps instead of twelve: the halves of the wave trade one value each (v_permlane32_swap), then every half reduces
// ONE value.  On return every lane holds both sums.
__device__ __forceinline__ void wave_reduce_fr2(Fr& a, Fr& b) {
    Fr lo, hi;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) {
        const auto s = __builtin_amdgcn_permlane32_swap(a.l[i], b.l[i], false, false);   // -> [a.lo, b.lo], [a.hi, b.hi]
        lo.l[i] = s[0];
        hi.l[i] = s[1];
    }
    Fr v = lo + hi;                     // lanes 0..31: partial sums of a, lanes 32..63: of b
    v = v + dpp_fr<0x111, 0xf>(v);
    v = v + dpp_fr<0x112, 0xf>(v);
    v = v + dpp_fr<0x114, 0xf>(v);
    v = v + dpp_fr<0x118, 0xf>(v);
    v = v + dpp_fr<0x142, 0xa>(v);      // lane 31: the sum of a, lane 63: the sum of b
    a = readlane_fr(v, 31);
    b = readlane_fr(v, 63);
}
// Sum over the workgroup; result valid in thread 0.  smem: (blockDim/64) Fr slots.
__device__ __forceinline__ Fr block_reduce_fr(Fr v, Fr* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63) >> 6;
    v = wave_reduce_fr(v);
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    Fr acc = Fr::zero();
    if (threadIdx.x == 0) {
        acc = smem[0];
        for (int w = 1; w < n_waves; ++w) acc = acc + smem[w];
    }
    __syncthreads();
    return acc;
}

// Two sums at once (one barrier set, two independent dependency chains for the scheduler to interleave).
__device__ __forceinline__ void block_reduce_fr2(Fr& a, Fr& b, Fr* smem /* 2 * n_waves */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63) >> 6;
    wave_reduce_fr2(a, b);
    if (lane == 0) { smem[2 * wave] = a; smem[2 * wave + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < n_waves; ++w) { a = a + smem[2 * w]; b = b + smem[2 * w + 1]; }
    }
    __syncthreads();
}

}  // namespace zk

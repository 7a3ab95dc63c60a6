// srs_kernels.hpp -- structured reference string generation on the device (gfx950).
//
// Replaces TrustedSetup::generate_powers_of_tau_in_g1 (kzg/src/trusted_setup.rs:25-35, with
// check_for_zero_and_one / generate_array_of_points kzg/src/utils.rs:19-40 over boolean_hypercube
// polynomial/src/utils.rs:141-157) and UnivariateKZG::generate_srs (kzg/src/univariate_kzg.rs:18-35):
// N independent fixed-base scalar multiplications G * s_i (byte-wide windows over a table of the generator's multiples), then one batched conversion to affine
// (the SRS is stored affine in HBM so that the commit path can use mixed additions).
#pragma once
#include "g1.hpp"
#include "mle_kernels.hpp"

namespace zk {

constexpr int SRS_BLOCK = 256;

// eq scalars of the multilinear SRS: s_i = prod_j (bit_j(i) ? tau_j : 1 - tau_j), bit 0 = MSB (variable 0)
static __global__ __launch_bounds__(SRS_BLOCK) void srs_eq_scalars_kernel(PtsArg tau, uint32_t n_vars,
                                                                   uint64_t* __restrict__ out) {
    const size_t n = (size_t)1 << n_vars;
    const size_t stride = (size_t)gridDim.x * SRS_BLOCK;
    const Fr one = Fr::one();
    for (size_t i = (size_t)blockIdx.x * SRS_BLOCK + threadIdx.x; i < n; i += stride) {
        Fr acc = one;
        for (uint32_t j = 0; j < n_vars; ++j) {
            Fr t = fr_from_pts(tau, j);
            if (!((i >> (n_vars - 1 - j)) & 1)) t = one - t;
            acc = acc * t;
        }
        store_fr(out, i, acc);
    }
}

// powers of tau: s_i = tau^i (tau.pow([i]) univariate_kzg.rs:26)
static __global__ __launch_bounds__(SRS_BLOCK) void srs_power_scalars_kernel(FrArg tau, size_t n,
                                                                      uint64_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * SRS_BLOCK;
    const Fr t = fr_from_arg(tau);
    for (size_t i = (size_t)blockIdx.x * SRS_BLOCK + threadIdx.x; i < n; i += stride) {
        Fr acc = Fr::one();
        bool started = false;
        for (int b = 63; b >= 0; --b) {
            if (started) acc = acc * acc;
            if ((i >> b) & 1) { acc = acc * t; started = true; }
        }
        store_fr(out, i, acc);
    }
}

__device__ __forceinline__ G1Affine g1_generator() {
    constexpr uint32_t gx[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                                 0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
    constexpr uint32_t gy[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                                 0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};
    G1Affine g;
#pragma unroll
    for (int i = 0; i < 12; ++i) { g.x.l[i] = gx[i]; g.y.l[i] = gy[i]; }
    return g;
}

// ---- windowed fixed base: 32 byte-wide windows of the scalar, table[w * 255 + d - 1] = d * 2^(8 w) * G -------------------
// 32 mixed additions per point instead of 255 doublings + ~127 additions.  The table depends on G only: one lane per window
// builds its 255 multiples by a chain of additions (XYZZ), srs_batch_affine_kernel turns them into affine points.
constexpr int SRS_WINDOWS = 32, SRS_DIGITS = 255;
static __global__ __launch_bounds__(64) void srs_gen_table_kernel(uint64_t* __restrict__ out_xyzz) {
    const uint32_t w = blockIdx.x * 64 + threadIdx.x;
    if (w >= SRS_WINDOWS) return;
    G1Xyzz base = G1Xyzz::identity();
    g1_madd(base, g1_generator(), false);
    for (uint32_t i = 0; i < 8 * w; ++i) base = g1_double(base);          // 2^(8 w) * G
    G1Xyzz acc = base;
    store_xyzz(out_xyzz, (size_t)w * SRS_DIGITS, acc);
    for (uint32_t d = 2; d <= SRS_DIGITS; ++d) {
        g1_add(acc, base);
        store_xyzz(out_xyzz, (size_t)w * SRS_DIGITS + d - 1, acc);
    }
}
// out[i] = G * scalars[i]  (Group::mul_bigint(point.into_bigint()) trusted_setup.rs:33) from the window table
static __global__ __launch_bounds__(SRS_BLOCK) void srs_fixed_base_window_kernel(const uint64_t* __restrict__ scalars, size_t n,
                                                                          const uint64_t* __restrict__ table_xy,
                                                                          uint64_t* __restrict__ out_xyzz) {
    const size_t i = (size_t)blockIdx.x * SRS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const Fr k = load_fr(scalars, i).from_mont();
    G1Xyzz acc = G1Xyzz::identity();
    for (int w = 0; w < SRS_WINDOWS; ++w) {
        uint32_t word = k.l[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) if (q == (w >> 2)) word = k.l[q];
        const uint32_t d = (word >> (8 * (w & 3))) & 255u;
        if (d) g1_madd(acc, load_affine(table_xy, (size_t)w * SRS_DIGITS + d - 1), false);
    }
    store_xyzz(out_xyzz, i, acc);
}

// a^(p-2) in Fq (Fermat)
__device__ __noinline__ Fq fq_inverse(Fq a) {
    Fq acc = Fq::one();
    for (int w = 11; w >= 0; --w) {
        uint32_t e = FqParams::p(0) - 2;
#pragma unroll
        for (int q = 0; q < 12; ++q) if (q == w) e = (q == 0) ? FqParams::p(0) - 2 : FqParams::p(q);
        for (int b = 31; b >= 0; --b) {
            acc = fq_sqr(acc);
            if ((e >> b) & 1) acc = fq_mul(acc, a);
        }
    }
    return acc;
}

// XYZZ -> affine (x = X/ZZ, y = Y/ZZZ) with Montgomery's batch-inversion trick over CHUNK points per lane.
constexpr int SRS_CHUNK = 8;
static __global__ __launch_bounds__(SRS_BLOCK) void srs_batch_affine_kernel(const uint64_t* __restrict__ in_xyzz, size_t n,
                                                                     uint64_t* __restrict__ out_xy,
                                                                     uint8_t* __restrict__ out_inf) {
    const size_t t = (size_t)blockIdx.x * SRS_BLOCK + threadIdx.x;
    const size_t base = t * SRS_CHUNK;
    if (base >= n) return;
    const int cnt = (int)((n - base) < (size_t)SRS_CHUNK ? (n - base) : SRS_CHUNK);
    Fq prefix[SRS_CHUNK];
    Fq acc = Fq::one();
    for (int k = 0; k < cnt; ++k) {
        prefix[k] = acc;
        Fq zz = load_fq(in_xyzz + 24 * (base + k) + 12), zzz = load_fq(in_xyzz + 24 * (base + k) + 18);
        if (!zz.is_zero()) acc = fq_mul(acc, fq_mul(zz, zzz));
    }
    Fq inv = fq_inverse(acc);
    for (int k = cnt - 1; k >= 0; --k) {
        G1Xyzz p = load_xyzz(in_xyzz, base + k);
        if (p.zz.is_zero()) {
            store_fq(out_xy + 12 * (base + k), Fq::zero());
            store_fq(out_xy + 12 * (base + k) + 6, Fq::zero());
            out_inf[base + k] = 1;
            continue;
        }
        Fq d = fq_mul(p.zz, p.zzz);
        Fq dinv = fq_mul(inv, prefix[k]);     // 1 / (zz * zzz) of this point
        inv = fq_mul(inv, d);
        store_fq(out_xy + 12 * (base + k), fq_mul(p.x, fq_mul(dinv, p.zzz)));
        store_fq(out_xy + 12 * (base + k) + 6, fq_mul(p.y, fq_mul(dinv, p.zz)));
        out_inf[base + k] = 0;
    }
}


// ---- folded SRS for MultilinearKZG::open ----------------------------------------------------------------
// out[j] = P[j] + P[j + h] for the affine SRS (first fold level), result XYZZ
static __global__ __launch_bounds__(SRS_BLOCK) void srs_fold_affine_kernel(const uint64_t* __restrict__ pts,
                                                                    const uint8_t* __restrict__ inf, size_t h,
                                                                    uint64_t* __restrict__ out_xyzz) {
    const size_t j = (size_t)blockIdx.x * SRS_BLOCK + threadIdx.x;
    if (j >= h) return;
    G1Xyzz acc = G1Xyzz::identity();
    if (!(inf && inf[j])) g1_madd(acc, load_affine(pts, j), false);
    if (!(inf && inf[j + h])) g1_madd(acc, load_affine(pts, j + h), false);
    store_xyzz(out_xyzz, j, acc);
}
// out[j] = in[j] + in[j + h] (further levels)
static __global__ __launch_bounds__(SRS_BLOCK) void srs_fold_xyzz_kernel(const uint64_t* __restrict__ in_xyzz, size_t h,
                                                                  uint64_t* __restrict__ out_xyzz) {
    const size_t j = (size_t)blockIdx.x * SRS_BLOCK + threadIdx.x;
    if (j >= h) return;
    G1Xyzz acc = load_xyzz(in_xyzz, j);
    g1_add(acc, load_xyzz(in_xyzz, j + h));
    store_xyzz(out_xyzz, j, acc);
}

}  // namespace zk

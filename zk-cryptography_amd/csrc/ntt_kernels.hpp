// ntt_kernels.hpp -- radix-2 number-theoretic transform over BLS12-381 Fr for gfx950.
//
// Replaces serial_fft (polynomial/src/utils.rs:281-315: bit-reversal permutation, then log2(n) in-place
// decimation-in-time butterfly stages with twiddle w_m^j, w_m = w^(n/2m)) as driven by
// Domain::{fft,ifft}_internal (polynomial/src/univariate/domain.rs:120-133) and
// UnivariateEval::multiply (polynomial/src/univariate/evaluation.rs:59-86).  Field arithmetic is exact, so
// any evaluation order yields the reference's values bit for bit.
//
// Passes over HBM (n x 32 bytes read + written each):
//   1. bit-reversal gather fused with the first NTT_TILE_LOG stages: a tile of 2^NTT_TILE_LOG consecutive
//      outputs of those stages depends only on the same consecutive (bit-reversed) inputs, so they run in LDS;
//   2. one pass per remaining stage (butterflies at distance >= the tile).
// Twiddles come from a table W[i] = w^i, i < n/2, built once per (size, direction) and kept in HBM.
#pragma once
#include "mle_kernels.hpp"

namespace zk {

constexpr int NTT_TILE_LOG = 10;               // 1024 elements = 32 KiB of LDS per workgroup
constexpr int NTT_TILE = 1 << NTT_TILE_LOG;

__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return __brev(x) >> (32 - bits); }

// W[i] = w^i from the repeated squares pw[k] = w^(2^k)
static __global__ __launch_bounds__(MLE_BLOCK) void ntt_twiddle_kernel(const uint64_t* __restrict__ pw, uint32_t log_half,
                                                                uint64_t* __restrict__ out) {
    const size_t n = (size_t)1 << log_half;
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t i = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; i < n; i += stride) {
        Fr acc = Fr::one();
        for (uint32_t k = 0; k < log_half; ++k)
            if ((i >> k) & 1) acc = acc * load_fr(pw, k);
        store_fr(out, i, acc);
    }
}

// pass 1: out[tile] = first min(log_n, NTT_TILE_LOG) stages applied to the bit-reversed input
static __global__ __launch_bounds__(MLE_BLOCK) void ntt_first_stages_kernel(const uint64_t* __restrict__ in,
                                                                     uint64_t* __restrict__ out, uint32_t log_n,
                                                                     const uint64_t* __restrict__ tw) {
    __shared__ Fr tab[NTT_TILE];
    const uint32_t n = 1u << log_n;
    const uint32_t tile = min((uint32_t)NTT_TILE, n);
    const uint32_t stages = min(log_n, (uint32_t)NTT_TILE_LOG);
    const uint32_t base = blockIdx.x * tile;
    for (uint32_t q = threadIdx.x; q < tile; q += MLE_BLOCK) tab[q] = load_fr(in, bitrev(base + q, log_n));
    __syncthreads();
    for (uint32_t s = 0; s < stages; ++s) {
        const uint32_t m = 1u << s;                 // butterfly distance
        const uint32_t tw_stride = n >> (s + 1);    // w_m^j = W[j * n/(2m)]
        for (uint32_t b = threadIdx.x; b < tile / 2; b += MLE_BLOCK) {
            const uint32_t j = b & (m - 1);
            const uint32_t i0 = ((b >> s) << (s + 1)) | j;
            Fr t = tab[i0 + m] * load_fr(tw, (size_t)j * tw_stride);
            Fr u = tab[i0];
            tab[i0 + m] = u - t;
            tab[i0] = u + t;
        }
        __syncthreads();
    }
    for (uint32_t q = threadIdx.x; q < tile; q += MLE_BLOCK) store_fr(out, base + q, tab[q]);
}

// one in-place stage with butterfly distance m = 2^s (s >= NTT_TILE_LOG)
static __global__ __launch_bounds__(MLE_BLOCK) void ntt_stage_kernel(uint64_t* __restrict__ data, uint32_t log_n, uint32_t s,
                                                              const uint64_t* __restrict__ tw) {
    const size_t half = (size_t)1 << (log_n - 1);
    const size_t m = (size_t)1 << s;
    const size_t tw_stride = (size_t)1 << (log_n - s - 1);
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t b = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; b < half; b += stride) {
        const size_t j = b & (m - 1);
        const size_t i0 = ((b >> s) << (s + 1)) | j;
        Fr t = load_fr(data, i0 + m) * load_fr(tw, j * tw_stride);
        Fr u = load_fr(data, i0);
        store_fr(data, i0 + m, u - t);
        store_fr(data, i0, u + t);
    }
}

// T consecutive stages s0 .. s0+T-1 (butterfly distances 2^s0 ..) in one pass.  Index i = hi | mid | lo with `mid` the T
// bits those stages pair up: for fixed (hi, lo) the 2^T elements form an independent sub-transform.  A workgroup
// stages 2^T mids x NTT_COLS consecutive `lo` values (512-byte contiguous pieces) in LDS, runs the T stages there
// and writes the tile back in place.
constexpr int NTT_MID_TILE_LOG = 10;              // 1024 elements = 32 KiB of LDS: several workgroups per CU hide the HBM latency
constexpr int NTT_MID_MAX = 6;                    // stages per pass; the tile is 2^T mids x (1024 >> T) consecutive lo values
static __global__ __launch_bounds__(MLE_BLOCK) void ntt_mid_stages_kernel(uint64_t* __restrict__ data, uint32_t log_n,
                                                                          uint32_t s0, uint32_t T,
                                                                          const uint64_t* __restrict__ tw) {
    __shared__ Fr tab[1 << NTT_MID_TILE_LOG];
    const uint32_t cols_log = NTT_MID_TILE_LOG - T;             // >= 4: pieces of >= 512 contiguous bytes
    const uint32_t cols = 1u << cols_log;
    const uint32_t lo_chunks = 1u << (s0 - cols_log);
    const size_t hi = blockIdx.x / lo_chunks;
    const uint32_t lo0 = (blockIdx.x % lo_chunks) << cols_log;
    const size_t base = (hi << (s0 + T)) | lo0;
    constexpr uint32_t tile = 1u << NTT_MID_TILE_LOG;
    for (uint32_t q = threadIdx.x; q < tile; q += MLE_BLOCK) {
        const uint32_t mid = q >> cols_log, c = q & (cols - 1);
        tab[q] = load_fr(data, base + ((size_t)mid << s0) + c);
    }
    __syncthreads();
    for (uint32_t t = 0; t < T; ++t) {
        const uint32_t s = s0 + t;
        const size_t tw_stride = (size_t)1 << (log_n - s - 1);
        for (uint32_t b = threadIdx.x; b < tile / 2; b += MLE_BLOCK) {
            const uint32_t c = b & (cols - 1), q = b >> cols_log;
            const uint32_t mid0 = ((q >> t) << (t + 1)) | (q & ((1u << t) - 1));
            const uint32_t i0 = (mid0 << cols_log) | c, i1 = i0 + (cols << t);
            const size_t j = ((size_t)(mid0 & ((1u << t) - 1)) << s0) | (lo0 + c);   // position inside the 2^s block
            Fr tt = tab[i1] * load_fr(tw, j * tw_stride);
            Fr u = tab[i0];
            tab[i1] = u - tt;
            tab[i0] = u + tt;
        }
        __syncthreads();
    }
    for (uint32_t q = threadIdx.x; q < tile; q += MLE_BLOCK) {
        const uint32_t mid = q >> cols_log, c = q & (cols - 1);
        store_fr(data, base + ((size_t)mid << s0) + c, tab[q]);
    }
}

// out[i] = a[i] * b[i]   (evaluation.rs:79-82)
static __global__ __launch_bounds__(MLE_BLOCK) void pointwise_mul_kernel(const uint64_t* __restrict__ a,
                                                                  const uint64_t* __restrict__ b, size_t n,
                                                                  uint64_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t i = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; i < n; i += stride)
        store_fr(out, i, load_fr(a, i) * load_fr(b, i));
}

}  // namespace zk

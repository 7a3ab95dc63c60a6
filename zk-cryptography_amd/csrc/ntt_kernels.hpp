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

// ---- transforms of >= 2^12 points: 2048-element tiles, every HBM access in >= 256-byte pieces ------------------------
// The radix-2 stages are grouped into passes; a pass keeps a tile of 2048 elements (64 KiB) in LDS for its T stages:
//   pass 1 (stages 0..7, bit reversal fused): a workgroup owns EIGHT 256-point sub-transforms whose output blocks differ in
//           their top three index bits -- the bit-reversed inputs of the eight then sit next to each other, so the gather
//           reads 256-byte pieces instead of one 32-byte element per 64 KiB;
//   pass p > 1 (stages s0 .. s0+T-1, T <= 7): index = hi | mid (T bits) | lo (s0 bits); a tile is 2^T mids x 2^(11-T)
//           consecutive lo values (>= 512-byte pieces).  The last pass writes to the destination (and folds the inverse
//           transform's 1/n into its last stage: u * c and v * (c w), one extra product per butterfly of one stage).
// Twiddles: the transform is bound by field products (n/2 log2 n of them, ~360 VALU instructions per butterfly), not by
// HBM, so what matters is that a twiddle costs one coalesced, prefetched load: every pass has its own table in tile order,
// T[t][ml][lo] = w^(((ml << s0) | lo) << (log_n - s0 - t - 1)), t < T, ml < 2^t, lo < 2^s0 (built once per size and
// direction from W); a lane loads the next stage's twiddles before it computes the current stage.
constexpr int NTT_BIG_TILE_LOG = 11;
constexpr int NTT_BIG_TILE = 1 << NTT_BIG_TILE_LOG;
constexpr int NTT_BIG_BLOCK = 512;
constexpr int NTT_FIRST_STAGES = 8;

// tw1[(1 << t) - 1 + j] = w^(j << (log_n - t - 1)), t < 8, j < 2^t
static __global__ __launch_bounds__(MLE_BLOCK) void ntt_first_table_kernel(const uint64_t* __restrict__ W, uint32_t log_n,
                                                                           uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * MLE_BLOCK + threadIdx.x;
    if (i >= (1u << NTT_FIRST_STAGES) - 1) return;
    const uint32_t t = 31 - __builtin_clz(i + 1), j = i + 1 - (1u << t);
    store_fr(out, i, load_fr(W, (size_t)j << (log_n - t - 1)));
}
// out[off(t) + (ml << s0) + lo] = W[((ml << s0) | lo) << (log_n - s0 - t - 1)], off(t) = (2^t - 1) 2^s0.  scale (nullable): the
// last stage's entries are multiplied by it (inverse transform)
static __global__ __launch_bounds__(MLE_BLOCK) void ntt_pass_table_kernel(const uint64_t* __restrict__ W, uint32_t log_n, uint32_t s0,
                                                                          uint32_t T, FrArg scale, uint32_t scaled,
                                                                          uint64_t* __restrict__ out) {
    const size_t total = (((size_t)1 << T) - 1) << s0;
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t i = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; i < total; i += stride) {
        const size_t q = (i >> s0) + 1;                       // 2^t + ml
        const uint32_t t = 63 - __builtin_clzll((unsigned long long)q);
        const size_t ml = q - ((size_t)1 << t), lo = i & (((size_t)1 << s0) - 1);
        Fr w = load_fr(W, ((ml << s0) | lo) << (log_n - s0 - t - 1));
        if (scaled && t == T - 1) w = w * fr_from_arg(scale);
        store_fr(out, i, w);
    }
}

// in: n_src <= n elements, zero beyond (coeffs.resize(size, F::zero()), domain.rs:109-110); in2 (nullable, n elements): the
// input is the element-wise product in * in2 (UnivariateEval::multiply's evaluation-form product, evaluation.rs:79-82,
// fused into the inverse transform's gather)
static __global__ __launch_bounds__(NTT_BIG_BLOCK) void ntt_first8_kernel(const uint64_t* __restrict__ in, size_t n_src,
                                                                          const uint64_t* __restrict__ in2, uint64_t* __restrict__ out,
                                                                          uint32_t log_n, const uint64_t* __restrict__ tw1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    Fr* tab = reinterpret_cast<Fr*>(zk_dyn_lds);
    const uint32_t g = blockIdx.x;
    for (uint32_t e = threadIdx.x; e < (uint32_t)NTT_BIG_TILE; e += NTT_BIG_BLOCK) {
        const uint32_t k = e & 7, q = e >> 3;                 // the eight sub-transforms' inputs are neighbours in memory
        const uint32_t o = (k << (log_n - 3)) | (g << 8) | q;
        const uint32_t src = bitrev(o, log_n);
        Fr v = src < n_src ? load_fr(in, src) : Fr::zero();
        if (in2) v = v * load_fr(in2, src);
        tab[k * 256 + q] = v;
    }
    __syncthreads();
    // stage 0: every twiddle is one
    for (uint32_t b = threadIdx.x; b < (uint32_t)NTT_BIG_TILE / 2; b += NTT_BIG_BLOCK) {
        const Fr u = tab[2 * b], v = tab[2 * b + 1];
        tab[2 * b] = u + v;
        tab[2 * b + 1] = u - v;
    }
    __syncthreads();
    const uint32_t b0 = threadIdx.x, b1 = threadIdx.x + NTT_BIG_BLOCK;      // this lane's two butterflies (bb = b & 127 is the same for both)
    const uint32_t bb = b0 & 127;
    Fr w = load_fr(tw1, 1 + (bb & 1));
    for (uint32_t t = 1; t < (uint32_t)NTT_FIRST_STAGES; ++t) {
        const uint32_t m = 1u << t, j = bb & (m - 1);
        const uint32_t i0 = ((bb >> t) << (t + 1)) | j;
        const Fr wc = w;
        if (t + 1 < (uint32_t)NTT_FIRST_STAGES) w = load_fr(tw1, (2 * m - 1) + (bb & (2 * m - 1)));   // next stage's, in flight during this one
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t idx0 = ((h ? b1 : b0) >> 7) * 256 + i0;
            const Fr tt = tab[idx0 + m] * wc;
            const Fr u = tab[idx0];
            tab[idx0 + m] = u - tt;
            tab[idx0] = u + tt;
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < (uint32_t)NTT_BIG_TILE; e += NTT_BIG_BLOCK) {
        const uint32_t k = e >> 8, q = e & 255;
        store_fr(out, ((size_t)k << (log_n - 3)) | ((size_t)g << 8) | q, tab[e]);
    }
}

template <bool LAST_SCALED>
static __global__ __launch_bounds__(NTT_BIG_BLOCK) void ntt_pass_kernel(const uint64_t* src, uint64_t* dst,   /* may alias: the middle passes run in place */
                                                                        uint32_t s0, uint32_t T, const uint64_t* __restrict__ tw,
                                                                        FrArg scale, size_t n_dst) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    Fr* tab = reinterpret_cast<Fr*>(zk_dyn_lds);
    const uint32_t cols_log = NTT_BIG_TILE_LOG - T, cols = 1u << cols_log;
    const uint32_t lo_chunks = 1u << (s0 - cols_log);
    const size_t hi = blockIdx.x / lo_chunks;
    const uint32_t lo0 = (blockIdx.x % lo_chunks) << cols_log;
    const size_t base = (hi << (s0 + T)) | lo0;
    for (uint32_t q = threadIdx.x; q < (uint32_t)NTT_BIG_TILE; q += NTT_BIG_BLOCK) {
        const uint32_t mid = q >> cols_log, c = q & (cols - 1);
        tab[q] = load_fr(src, base + ((size_t)mid << s0) + c);
    }
    // this lane's two butterflies per stage: b = threadIdx.x and threadIdx.x + 512 -> (c, q = b >> cols_log)
    const uint32_t c = threadIdx.x & (cols - 1);
    const uint32_t q0 = threadIdx.x >> cols_log, q1 = (threadIdx.x + NTT_BIG_BLOCK) >> cols_log;
    auto tw_index = [&](uint32_t t, uint32_t q) -> size_t {
        const uint32_t ml = q & ((1u << t) - 1);
        return ((((size_t)1 << t) - 1) << s0) + ((size_t)ml << s0) + lo0 + c;
    };
    Fr w0 = load_fr(tw, tw_index(0, q0)), w1 = load_fr(tw, tw_index(0, q1));
    __syncthreads();
    for (uint32_t t = 0; t < T; ++t) {
        const Fr wa = w0, wb = w1;
        if (t + 1 < T) { w0 = load_fr(tw, tw_index(t + 1, q0)); w1 = load_fr(tw, tw_index(t + 1, q1)); }
        const bool last = LAST_SCALED && t + 1 == T;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t q = h ? q1 : q0;
            const uint32_t mid0 = ((q >> t) << (t + 1)) | (q & ((1u << t) - 1));
            const uint32_t i0 = (mid0 << cols_log) | c, i1 = i0 + (cols << t);
            const Fr tt = tab[i1] * (h ? wb : wa);
            Fr u = tab[i0];
            if (last) u = u * fr_from_arg(scale);
            tab[i1] = u - tt;
            tab[i0] = u + tt;
        }
        __syncthreads();
    }
    for (uint32_t q = threadIdx.x; q < (uint32_t)NTT_BIG_TILE; q += NTT_BIG_BLOCK) {
        const uint32_t mid = q >> cols_log, cc = q & (cols - 1);
        const size_t o = base + ((size_t)mid << s0) + cc;
        if (o < n_dst) store_fr(dst, o, tab[q]);              // a product keeps len_a + len_b - 1 coefficients (evaluation.rs:85)
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

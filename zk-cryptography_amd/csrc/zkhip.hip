// zkhip.hip -- C-ABI entry points (include/zkhip.h) and host-side launch logic.
// gfx950 only.  No CPU fallback: every entry point launches HIP kernels or fails.
#include "../../include/zkhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "composed_kernels.hpp"
#include "host_fr.hpp"
#include "host_g1.hpp"
#include "mle_kernels.hpp"
#include "msm_kernels.hpp"
#include "srs_kernels.hpp"
#include "sumcheck_kernels.hpp"

using namespace zk;

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
static int g_last_hip = 0;   // last HIP error seen without a context (ctx_create)
extern "C" int zkhip_version(void) { return 1; }

extern "C" const char* zkhip_status_string(int s) {
    switch (s) {
        case ZKHIP_OK: return "ok";
        case ZKHIP_ERR_HIP: return "HIP runtime error";
        case ZKHIP_ERR_SHAPE: return "shape assertion failed";
        case ZKHIP_ERR_INDEX: return "index out of bounds";
        case ZKHIP_ERR_ARG: return "invalid argument";
        case ZKHIP_ERR_NOMEM: return "out of memory";
        default: return "unknown";
    }
}

extern "C" int zkhip_ctx_create(zkhip_ctx** out, int device, void* stream) {
    if (!out) return ZKHIP_ERR_ARG;
    int count = 0;
    hipError_t e0 = hipGetDeviceCount(&count);
    if (e0 != hipSuccess || device < 0 || device >= count) { g_last_hip = (int)e0; return ZKHIP_ERR_HIP; }
    zkhip_ctx* c = new zkhip_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return ZKHIP_ERR_HIP; }
    c->stream = (hipStream_t)stream;   // NULL = the device's default (null) stream, ordered with everything
    c->own_stream = false;
    if (hipHostMalloc(&c->h_pinned, ZK_PINNED_BYTES, hipHostMallocDefault) != hipSuccess) { delete c; return ZKHIP_ERR_HIP; }
    if (hipMalloc(&c->d_small, ZK_SMALL_BYTES) != hipSuccess) { delete c; return ZKHIP_ERR_NOMEM; }
    *out = c;
    return ZKHIP_OK;
}

extern "C" int zkhip_ctx_destroy(zkhip_ctx* c) {
    if (!c) return ZKHIP_ERR_ARG;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto& e : c->prof_events) { hipEventDestroy(e.start); hipEventDestroy(e.stop); }
    if (c->d_ws) hipFree(c->d_ws);
    if (c->d_small) hipFree(c->d_small);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return ZKHIP_OK;
}

extern "C" int zkhip_ctx_set_stream(zkhip_ctx* c, void* stream) {
    if (!c) return ZKHIP_ERR_ARG;
    c->stream = (hipStream_t)stream;
    return ZKHIP_OK;
}
extern "C" int zkhip_ctx_synchronize(zkhip_ctx* c) {
    if (!c) return ZKHIP_ERR_ARG;
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZKHIP_OK;
}
extern "C" int zkhip_last_hip_error(zkhip_ctx* c) { return c ? c->last_hip : g_last_hip; }

extern "C" int zkhip_malloc(zkhip_ctx* c, void** d_ptr, size_t bytes) {
    if (!c || !d_ptr) return ZKHIP_ERR_ARG;
    hipSetDevice(c->device);
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) { c->last_hip = (int)e; return ZKHIP_ERR_NOMEM; }
    return ZKHIP_OK;
}
extern "C" int zkhip_free(zkhip_ctx* c, void* d_ptr) {
    if (!c) return ZKHIP_ERR_ARG;
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    ZK_HIP(c, hipFree(d_ptr));
    return ZKHIP_OK;
}
extern "C" int zkhip_memcpy_h2d(zkhip_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    if (!c) return ZKHIP_ERR_ARG;
    ZK_HIP(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZKHIP_OK;
}
extern "C" int zkhip_memcpy_d2h(zkhip_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    if (!c) return ZKHIP_ERR_ARG;
    ZK_HIP(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZKHIP_OK;
}

extern "C" int zkhip_profile_enable(zkhip_ctx* c, int enable) {
    if (!c) return ZKHIP_ERR_ARG;
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    c->profiling = enable != 0;
    c->prof_used = 0;
    c->prof_records.clear();
    return ZKHIP_OK;
}
extern "C" int zkhip_profile_read(zkhip_ctx* c, const char* kernel, double* total_ms, uint64_t* launches, double* bytes) {
    if (!c || !kernel) return ZKHIP_ERR_ARG;
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    double ms = 0, by = 0;
    uint64_t cnt = 0;
    for (auto& r : c->prof_records) {
        if (std::strcmp(r.name, kernel) != 0) continue;
        float t = 0;
        ZK_HIP(c, hipEventElapsedTime(&t, c->prof_events[r.event].start, c->prof_events[r.event].stop));
        ms += t; by += r.bytes; ++cnt;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = cnt;
    if (bytes) *bytes = by;
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------
static inline uint32_t log2_exact(size_t n) {
    uint32_t k = 0;
    while (((size_t)1 << k) < n) ++k;
    return k;
}
static inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }

// fold variable var_index of an n-entry table; r on device. with_sums -> partials (returns grid size in *np)
static int launch_fold(zkhip_ctx* c, const uint64_t* d_in, size_t n, const uint64_t* d_r, uint32_t var_index,
                       uint64_t* d_out, bool with_sums, uint64_t* d_partials, uint32_t* np) {
    const size_t n_out = n / 2;
    const uint32_t log_half = log2_exact(n) - 1 - var_index;
    const int grid = mle_grid((n_out + 1) / 2);
    ProfScope ps(c, with_sums ? "fold_sums" : "fold", 48.0 * (double)n);
    if (with_sums)
        hipLaunchKernelGGL(fold_kernel<true>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_in, d_out, n_out, log_half,
                           d_r, d_partials);
    else
        hipLaunchKernelGGL(fold_kernel<false>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_in, d_out, n_out, log_half,
                           d_r, d_partials);
    if (np) *np = (uint32_t)grid;
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// Multilinear
// ---------------------------------------------------------------------------------------
extern "C" int zkhip_mle_partial_evaluation(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_r,
                                            const uint64_t* d_r, uint32_t var_index, uint64_t* d_out) {
    if (!c || !d_evals || !d_out || (!h_r == !d_r)) return ZKHIP_ERR_ARG;
    if (!is_pow2(n) || n < 2) return ZKHIP_ERR_SHAPE;                 // utils.rs:30  (and Multilinear::new :16-20)
    if (!((size_t)var_index < n / 2)) return ZKHIP_ERR_SHAPE;          // utils.rs:31-34
    if (var_index >= log2_exact(n)) return ZKHIP_ERR_SHAPE;            // reference would return an empty table
    ZK_TRY(c->activate());
    if (h_r) {
        uint64_t* slot = c->small_u64(ZK_SMALL_R);
        std::memcpy(c->pinned_u64(ZK_PIN_R), h_r, 32);
        ZK_HIP(c, hipMemcpyAsync(slot, c->pinned_u64(ZK_PIN_R), 32, hipMemcpyHostToDevice, c->stream));
        d_r = slot;
    }
    return launch_fold(c, d_evals, n, d_r, var_index, d_out, false, nullptr, nullptr);
}

// Successive folds of variable 0 (or of h_var_indices) with points already on the device.
static int fold_chain(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* d_pts,
                      const uint32_t* var_indices, size_t n_pts, uint64_t* d_out) {
    // ping-pong buffers: A holds n/2, B holds n/4
    if (n_pts == 0) {
        ZK_HIP(c, hipMemcpyAsync(d_out, d_evals, n * 32, hipMemcpyDeviceToDevice, c->stream));
        return ZKHIP_OK;
    }
    const size_t need = (n / 2 + n / 4 + 8) * 32;
    ZK_TRY(c->reserve_ws(need));
    uint64_t* A = (uint64_t*)c->d_ws;
    uint64_t* B = A + 4 * (n / 2);
    const uint64_t* cur = d_evals;
    size_t cn = n;
    for (size_t p = 0; p < n_pts; ++p) {
        const uint32_t k = var_indices ? var_indices[p] : 0;
        if (cn < 2 || !((size_t)k < cn / 2) || k >= log2_exact(cn)) return ZKHIP_ERR_SHAPE;
        const bool all_zero_tail = !var_indices || [&] {
            for (size_t q = p; q < n_pts; ++q) if (var_indices[q]) return false;
            return true;
        }();
        if (all_zero_tail && cn <= (size_t)TAIL_N) {
            // finish every remaining fold inside one workgroup
            ProfScope ps(c, "fold_tail", 0.0);
            ZK_TRY(c->allow_big_lds((const void*)fold_tail_kernel, TAIL_LDS_BYTES));
            hipLaunchKernelGGL(fold_tail_kernel, dim3(1), dim3(MLE_BLOCK), TAIL_LDS_BYTES, c->stream, cur, (uint32_t)cn,
                               d_pts + 4 * p, (uint32_t)(n_pts - p), d_out);
            ZK_HIP(c, hipGetLastError());
            return ZKHIP_OK;
        }
        const bool last = (p + 1 == n_pts);
        uint64_t* dst = last ? d_out : ((p & 1) ? B : A);
        ZK_TRY(launch_fold(c, cur, cn, d_pts + 4 * p, k, dst, false, nullptr, nullptr));
        cur = dst;
        cn /= 2;
    }
    return ZKHIP_OK;
}

extern "C" int zkhip_mle_partial_evaluations(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_pts,
                                             const uint32_t* h_var_indices, size_t n_pts, uint64_t* d_out) {
    if (!c || !d_evals || !d_out || (n_pts && (!h_pts || !h_var_indices))) return ZKHIP_ERR_ARG;
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;
    if (n_pts > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    uint64_t* d_pts = c->small_u64(ZK_SMALL_PTS);
    if (n_pts) {
        std::memcpy(c->pinned_u64(ZK_PIN_PTS), h_pts, 32 * n_pts);
        ZK_HIP(c, hipMemcpyAsync(d_pts, c->pinned_u64(ZK_PIN_PTS), 32 * n_pts, hipMemcpyHostToDevice, c->stream));
    }
    return fold_chain(c, d_evals, n, d_pts, h_var_indices, n_pts, d_out);
}

extern "C" int zkhip_mle_evaluation(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_pts,
                                    size_t n_pts, uint64_t* h_out) {
    if (!c || !d_evals || !h_out || (n_pts && !h_pts)) return ZKHIP_ERR_ARG;
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;
    if (n_pts != log2_exact(n)) return ZKHIP_ERR_SHAPE;                // assert_eq! evaluation_form.rs:163-167
    ZK_TRY(c->activate());
    uint64_t* d_pts = c->small_u64(ZK_SMALL_PTS);
    uint64_t* d_res = c->small_u64(ZK_SMALL_RES);
    if (n_pts) {
        std::memcpy(c->pinned_u64(ZK_PIN_PTS), h_pts, 32 * n_pts);
        ZK_HIP(c, hipMemcpyAsync(d_pts, c->pinned_u64(ZK_PIN_PTS), 32 * n_pts, hipMemcpyHostToDevice, c->stream));
    }
    ZK_TRY(fold_chain(c, d_evals, n, d_pts, nullptr, n_pts, d_res));
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_res, 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_out, c->pinned_u64(ZK_PIN_RES), 32);
    return ZKHIP_OK;
}

extern "C" int zkhip_mle_half_sums(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint64_t* h_out) {
    if (!c || !d_evals || !h_out) return ZKHIP_ERR_ARG;
    if (n == 0) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    uint64_t* d_partials = c->small_u64(ZK_SMALL_PARTIALS);
    uint64_t* d_res = c->small_u64(ZK_SMALL_RES);
    const int grid = mle_grid((n + 1) / 2);
    {
        ProfScope ps(c, "half_sums", 32.0 * (double)n);
        hipLaunchKernelGGL(half_sums_kernel, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_evals, n, d_partials);
    }
    hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_partials, (uint32_t)grid, d_res);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_res, 96, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_out, c->pinned_u64(ZK_PIN_RES), 96);
    return ZKHIP_OK;
}

static int distinct(zkhip_ctx* c, bool mul, const uint64_t* d_a, size_t na, const uint64_t* d_b, size_t nb,
                    uint64_t* d_out) {
    if (!c || !d_a || !d_b || !d_out) return ZKHIP_ERR_ARG;
    if (!is_pow2(na * nb)) return ZKHIP_ERR_SHAPE;   // Self::new(new_evaluations) asserts a power of two
    ZK_TRY(c->activate());
    const size_t n_out = na * nb;
    const int grid = mle_grid(n_out);
    if (mul)
        hipLaunchKernelGGL(distinct_kernel<true>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_a, d_b, nb, n_out, d_out);
    else
        hipLaunchKernelGGL(distinct_kernel<false>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_a, d_b, nb, n_out, d_out);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}
extern "C" int zkhip_mle_add_distinct(zkhip_ctx* c, const uint64_t* d_a, size_t na, const uint64_t* d_b, size_t nb,
                                      uint64_t* d_out) { return distinct(c, false, d_a, na, d_b, nb, d_out); }
extern "C" int zkhip_mle_mul_distinct(zkhip_ctx* c, const uint64_t* d_a, size_t na, const uint64_t* d_b, size_t nb,
                                      uint64_t* d_out) { return distinct(c, true, d_a, na, d_b, nb, d_out); }

extern "C" int zkhip_mle_elementwise(zkhip_ctx* c, int op, const uint64_t* d_a, const uint64_t* d_b,
                                     const uint64_t* h_scalar, size_t n, uint64_t* d_out) {
    if (!c || !d_a || !d_out || op < 0 || op > 2) return ZKHIP_ERR_ARG;
    if (op == 2 ? !h_scalar : !d_b) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    const int grid = mle_grid(n);
    if (op == 2) {
        uint64_t* slot = c->small_u64(ZK_SMALL_R);
        std::memcpy(c->pinned_u64(ZK_PIN_R), h_scalar, 32);
        ZK_HIP(c, hipMemcpyAsync(slot, c->pinned_u64(ZK_PIN_R), 32, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(elementwise_kernel<2>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_a, slot, n, d_out);
    } else if (op == 0) {
        hipLaunchKernelGGL(elementwise_kernel<0>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_a, d_b, n, d_out);
    } else {
        hipLaunchKernelGGL(elementwise_kernel<1>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_a, d_b, n, d_out);
    }
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

extern "C" int zkhip_mle_to_bytes(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint8_t* d_out_bytes) {
    if (!c || !d_evals || !d_out_bytes) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    hipLaunchKernelGGL(to_bytes_kernel, dim3(mle_grid(n)), dim3(MLE_BLOCK), 0, c->stream, d_evals, n,
                       (uint32_t*)d_out_bytes);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// basic sumcheck prover
// ---------------------------------------------------------------------------------------
extern "C" int zkhip_sumcheck_prove(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_claimed_sum,
                                    const uint64_t* h_first_half_sums, uint64_t* h_sum, uint64_t* h_round_polys,
                                    uint64_t* h_challenges) {
    if (!c || !d_evals || !h_sum) return ZKHIP_ERR_ARG;
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;   // Multilinear::new evaluation_form.rs:16-20
    const uint32_t n_vars = log2_exact(n);
    if (n_vars > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    if (n_vars && (!h_round_polys || !h_challenges)) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    if (n == 1) {   // no rounds: nothing is proven; report the sum the transcript would have absorbed
        if (h_claimed_sum) { std::memcpy(h_sum, h_claimed_sum, 32); return ZKHIP_OK; }
        ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_evals, 32, hipMemcpyDeviceToHost, c->stream));
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        std::memcpy(h_sum, c->pinned_u64(ZK_PIN_RES), 32);
        return ZKHIP_OK;
    }
    ZK_TRY(c->reserve_ws((n / 2 + n / 4 + 8) * 32));
    uint64_t* A = (uint64_t*)c->d_ws;
    uint64_t* B = A + 4 * (n / 2);
    SumcheckDev* st = (SumcheckDev*)c->small_u64(ZK_SMALL_STATE);
    uint64_t* d_partials = c->small_u64(ZK_SMALL_PARTIALS);
    uint64_t* d_rp = c->small_u64(ZK_SMALL_ROUNDPOLYS);
    uint64_t* d_ch = c->small_u64(ZK_SMALL_CHALLENGES);
    uint64_t* d_fin = c->small_u64(ZK_SMALL_RES);

    const uint64_t* cur = d_evals;
    size_t cn = n;
    uint32_t round = 0, first = 1, np = 0;
    if (h_claimed_sum) {   // prove(&self) absorbs self.sum, whatever the caller put there (sumcheck.rs:33-35)
        std::memcpy(c->pinned_u64(ZK_PIN_R), h_claimed_sum, 32);
        ZK_HIP(c, hipMemcpyAsync(st->sum, c->pinned_u64(ZK_PIN_R), 32, hipMemcpyHostToDevice, c->stream));
        first = 2;
    }
    if (cn > (size_t)TAIL_N) {
        if (h_first_half_sums) {   // poly_sum() already streamed the table once: reuse its two half sums
            std::memcpy(c->pinned_u64(ZK_PIN_RES), h_first_half_sums, 64);
            ZK_HIP(c, hipMemcpyAsync(d_partials, c->pinned_u64(ZK_PIN_RES), 64, hipMemcpyHostToDevice, c->stream));
            np = 1;
        } else {
            const int grid = mle_grid((cn + 1) / 2);
            ProfScope ps(c, "half_sums", 32.0 * (double)cn);
            hipLaunchKernelGGL(half_sums_kernel, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, cur, cn, d_partials);
            np = (uint32_t)grid;
        }
        while (cn > (size_t)TAIL_N) {
            hipLaunchKernelGGL(sumcheck_round_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_partials, np, st, round,
                               first, d_rp, d_ch);
            first = 0;
            uint64_t* dst = (round & 1) ? B : A;
            const bool next_is_tail = (cn / 2) <= (size_t)TAIL_N;
            ZK_TRY(launch_fold(c, cur, cn, d_ch + 4 * round, 0, dst, !next_is_tail, d_partials, &np));
            cur = dst;
            cn /= 2;
            ++round;
        }
    }
    ZK_TRY(c->allow_big_lds((const void*)sumcheck_tail_kernel, TAIL_LDS_BYTES));
    hipLaunchKernelGGL(sumcheck_tail_kernel, dim3(1), dim3(MLE_BLOCK), TAIL_LDS_BYTES, c->stream, cur, (uint32_t)cn, st, round, first,
                       d_rp, d_ch, d_fin);
    ZK_HIP(c, hipGetLastError());
    // results -> host
    uint64_t* pin = c->pinned_u64(ZK_PIN_PROOF);
    ZK_HIP(c, hipMemcpyAsync(pin, st->sum, 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipMemcpyAsync(pin + 4, d_rp, 64 * (size_t)n_vars, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipMemcpyAsync(pin + 4 + 8 * ZK_MAX_ROUNDS, d_ch, 32 * (size_t)n_vars, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_sum, pin, 32);
    std::memcpy(h_round_polys, pin + 4, 64 * (size_t)n_vars);
    std::memcpy(h_challenges, pin + 4 + 8 * ZK_MAX_ROUNDS, 32 * (size_t)n_vars);
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// composed / multi-composed sumcheck provers
// ---------------------------------------------------------------------------------------
template <int K>
static void launch_product_sum(zkhip_ctx* c, const TablePtrs& tp, size_t n, int grid, uint64_t* partials) {
    hipLaunchKernelGGL(product_sum_kernel<K>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, n, partials);
}
template <int K>
static void launch_round(zkhip_ctx* c, bool fold, const TablePtrs& tp, size_t n, const uint64_t* r, uint32_t rec,
                         uint32_t rec_off, uint64_t* partials, int grid) {
    if (fold)
        hipLaunchKernelGGL((composed_round_kernel<K, true>), dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, n, r, rec, rec_off, partials);
    else
        hipLaunchKernelGGL((composed_round_kernel<K, false>), dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, n, r, rec, rec_off, partials);
}
#define ZK_DISPATCH_K(k, CALL)                 \
    switch (k) {                               \
        case 1: CALL(1); break;                \
        case 2: CALL(2); break;                \
        case 3: CALL(3); break;                \
        case 4: CALL(4); break;                \
        case 5: CALL(5); break;                \
        default: return ZKHIP_ERR_ARG;         \
    }

static int product_sums(zkhip_ctx* c, const uint64_t* const* ptrs, const uint32_t* term_sizes, uint32_t n_terms, size_t n,
                        uint64_t* h_sum) {
    if (!c || !ptrs || !term_sizes || !h_sum) return ZKHIP_ERR_ARG;
    if (n == 0 || n_terms == 0 || n_terms > CMP_MAX_TERMS) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    uint64_t* d_partials = c->small_u64(ZK_SMALL_PARTIALS);
    uint64_t* d_res = c->small_u64(ZK_SMALL_RES);
    const int grid = mle_grid(n);
    size_t off = 0;
    for (uint32_t p = 0; p < n_terms; ++p) {
        TablePtrs tp = {};
        for (uint32_t q = 0; q < term_sizes[p] && q < CMP_MAX_K; ++q) tp.in[q] = ptrs[off + q];
#define CALL(KK) launch_product_sum<KK>(c, tp, n, grid, d_partials)
        ZK_DISPATCH_K(term_sizes[p], CALL)
#undef CALL
        hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_partials, (uint32_t)grid, d_res, p ? 1u : 0u);
        off += term_sizes[p];
    }
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_res, 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_sum, c->pinned_u64(ZK_PIN_RES), 32);
    return ZKHIP_OK;
}

extern "C" int zkhip_composed_sum(zkhip_ctx* c, const uint64_t* const* ptrs, uint32_t k, size_t n, uint64_t* h_sum) {
    return product_sums(c, ptrs, &k, 1, n, h_sum);
}
extern "C" int zkhip_multi_composed_sum(zkhip_ctx* c, const uint64_t* const* ptrs, const uint32_t* term_sizes,
                                        uint32_t n_terms, size_t n, uint64_t* h_sum) {
    return product_sums(c, ptrs, term_sizes, n_terms, n, h_sum);
}

// shared driver.  multi = 0: ComposedSumcheck (one term).  first_mode: see composed_transcript_kernel.
static int composed_prove_impl(zkhip_ctx* c, const uint64_t* const* ptrs, const uint32_t* term_sizes, uint32_t n_terms,
                               size_t n, int multi, const uint64_t* h_sum, int partial, uint32_t* h_lens,
                               uint64_t* h_round_polys, uint64_t* h_challenges) {
    if (!c || !ptrs || !term_sizes) return ZKHIP_ERR_ARG;
    if (n_terms == 0 || n_terms > CMP_MAX_TERMS) return ZKHIP_ERR_ARG;
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;
    const uint32_t n_vars = log2_exact(n);
    if (n_vars > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    if (n_vars == 0) return ZKHIP_OK;   // `for _ in 0..n_vars` never runs
    if (!h_round_polys || !h_challenges || (multi && (!h_sum || !h_lens))) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    ComposedMeta meta = {};
    meta.n_terms = n_terms;
    meta.multi = (uint32_t)multi;
    uint32_t total = 0, rec = 0;
    for (uint32_t p = 0; p < n_terms; ++p) {
        if (term_sizes[p] < 1 || term_sizes[p] > CMP_MAX_K) return ZKHIP_ERR_ARG;
        meta.k[p] = term_sizes[p];
        meta.rec_off[p] = rec;
        rec += term_sizes[p] + 1;
        total += term_sizes[p];
    }
    meta.rec = rec;
    if (rec > CMP_MAX_REC) return ZKHIP_ERR_ARG;
    // workspace: per table a ping (n/2) and a pong (n/4) buffer, then the state
    const size_t per_table = (n / 2 + n / 4 + 2) * 32;
    const size_t state_off = (total * per_table + 255) & ~(size_t)255;
    const size_t bytes_off = state_off + ((sizeof(ComposedDev) + 255) & ~(size_t)255);
    ZK_TRY(c->reserve_ws(bytes_off + (multi && !partial ? 32 * n : 0)));
    char* ws = (char*)c->d_ws;
    ComposedDev* st = (ComposedDev*)(ws + state_off);
    uint64_t* d_partials = c->small_u64(ZK_SMALL_PARTIALS);
    uint64_t* d_rp = c->small_u64(ZK_SMALL_ROUNDPOLYS);
    uint64_t* d_ch = c->small_u64(ZK_SMALL_CHALLENGES);

    uint32_t first = 1;
    if (multi) {
        // interpolation matrices for the degrees in use
        std::vector<uint64_t> mats((CMP_MAX_K + 1) * (CMP_MAX_K + 1) * (CMP_MAX_K + 1) * 4, 0);
        for (uint32_t p = 0; p < n_terms; ++p) {
            const int d = (int)term_sizes[p];
            std::vector<zkhost::Fr> m = zkhost::interpolation_matrix(d);
            std::memcpy(&mats[(size_t)d * (CMP_MAX_K + 1) * (CMP_MAX_K + 1) * 4], m.data(), m.size() * 32);
        }
        ZK_HIP(c, hipMemcpyAsync(st->interp, mats.data(), mats.size() * 8, hipMemcpyHostToDevice, c->stream));
        ZK_HIP(c, hipMemcpyAsync(st->sum, h_sum, 32, hipMemcpyHostToDevice, c->stream));
        if (!partial) {
            // prove(): transcript.commit(&composed_poly_to_bytes(&poly)) first (multi_composed_sumcheck.rs:51-53).
            // The GPU produces the canonical big-endian bytes, the host hashes the (inherently sequential) stream.
            uint8_t* d_bytes = (uint8_t*)(ws + bytes_off);
            std::vector<uint8_t> h_bytes(32 * n);
            zkhost::Sha256 sha;
            for (uint32_t q = 0; q < total; ++q) {
                hipLaunchKernelGGL(to_bytes_kernel, dim3(mle_grid(n)), dim3(MLE_BLOCK), 0, c->stream, ptrs[q], n, (uint32_t*)d_bytes);
                ZK_HIP(c, hipMemcpyAsync(h_bytes.data(), d_bytes, 32 * n, hipMemcpyDeviceToHost, c->stream));
                ZK_HIP(c, hipStreamSynchronize(c->stream));
                sha.update(h_bytes.data(), 32 * n);
            }
            Sha256State hs = {};
            std::memcpy(hs.h, sha.h, 32);
            const uint32_t fill = (uint32_t)(sha.len % 64);
            for (uint32_t i = 0; i < fill / 4; ++i)
                hs.buf[i] = ((uint32_t)sha.buf[4 * i] << 24) | ((uint32_t)sha.buf[4 * i + 1] << 16) | ((uint32_t)sha.buf[4 * i + 2] << 8) | sha.buf[4 * i + 3];
            hs.fill = fill;
            hs.len = sha.len;
            ZK_HIP(c, hipMemcpyAsync(&st->transcript, &hs, sizeof(hs), hipMemcpyHostToDevice, c->stream));
            ZK_HIP(c, hipStreamSynchronize(c->stream));   // hs / mats are stack/heap temporaries
            first = 2;
        } else {
            ZK_HIP(c, hipStreamSynchronize(c->stream));
        }
    }

    // current table pointers
    std::vector<const uint64_t*> cur(ptrs, ptrs + total);
    size_t cn = n;
    for (uint32_t round = 0; round < n_vars; ++round) {
        const bool fold = round > 0;
        const size_t work = fold ? cn / 4 : cn / 2;
        const int grid = mle_grid(work ? work : 1);
        uint32_t off = 0;
        for (uint32_t p = 0; p < n_terms; ++p) {
            TablePtrs tp = {};
            // ping-pong: folds happen in rounds 1, 2, ...; round r writes n >> r entries.  Odd rounds use the
            // n/2-entry buffer, even rounds the n/4-entry one.
            for (uint32_t q = 0; q < term_sizes[p]; ++q) {
                tp.in[q] = cur[off + q];
                char* base = ws + (size_t)(off + q) * per_table;
                tp.out[q] = (uint64_t*)((round & 1) ? base : base + (n / 2 + 1) * 32);
            }
            ProfScope ps(c, "composed_round", 0.0);
#define CALL(KK) launch_round<KK>(c, fold, tp, cn, fold ? d_ch + 4 * (round - 1) : nullptr, meta.rec, meta.rec_off[p], d_partials, grid)
            ZK_DISPATCH_K(term_sizes[p], CALL)
#undef CALL
            if (fold) for (uint32_t q = 0; q < term_sizes[p]; ++q) cur[off + q] = tp.out[q];
            off += term_sizes[p];
        }
        if (fold) cn /= 2;
        hipLaunchKernelGGL(composed_transcript_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_partials, (uint32_t)grid, meta, st,
                           round, first, d_rp, d_ch);
        first = 0;
    }
    ZK_HIP(c, hipGetLastError());
    std::vector<uint64_t> h_rp(64 * (size_t)n_vars);
    ZK_HIP(c, hipMemcpyAsync(h_rp.data(), d_rp, 64 * 8 * (size_t)n_vars, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipMemcpyAsync(h_challenges, d_ch, 32 * (size_t)n_vars, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t r = 0; r < n_vars; ++r) {
        if (!multi) {
            std::memcpy(h_round_polys + (size_t)r * (term_sizes[0] + 1) * 4, &h_rp[64 * r], (term_sizes[0] + 1) * 32);
        } else {
            h_lens[r] = (uint32_t)h_rp[64 * r];
            std::memcpy(h_round_polys + (size_t)r * CMP_MAX_MONO * 8, &h_rp[64 * r + 8], CMP_MAX_MONO * 64);
        }
    }
    return ZKHIP_OK;
}

extern "C" int zkhip_composed_prove(zkhip_ctx* c, const uint64_t* const* ptrs, uint32_t k, size_t n, uint64_t* h_round_polys,
                                    uint64_t* h_challenges) {
    return composed_prove_impl(c, ptrs, &k, 1, n, 0, nullptr, 1, nullptr, h_round_polys, h_challenges);
}
extern "C" int zkhip_multi_composed_prove(zkhip_ctx* c, const uint64_t* const* ptrs, const uint32_t* term_sizes,
                                          uint32_t n_terms, size_t n, const uint64_t* h_sum, int partial,
                                          uint32_t* h_lens, uint64_t* h_round_polys, uint64_t* h_challenges) {
    return composed_prove_impl(c, ptrs, term_sizes, n_terms, n, 1, h_sum, partial, h_lens, h_round_polys, h_challenges);
}

// ---------------------------------------------------------------------------------------
// KZG commit (MSM)
// ---------------------------------------------------------------------------------------
static MsmPlan msm_plan(size_t n) {
    uint32_t lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
    uint32_t c = lg > 2 ? lg - 2 : 4;            // about 8 points per bucket per window on small inputs
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    MsmPlan pl;
    pl.c = c;
    pl.n_windows = (256 + c - 1) / c;
    pl.nb = 1u << (c - 1);
    pl.ns = pl.nb / MSM_SEG;
    pl.n_bits = c - 1 - MSM_SEG_LOG;
    pl.n_terms = 1 + pl.n_bits;
    return pl;
}

extern "C" int zkhip_kzg_commit(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf,
                                size_t n_points, const uint64_t* d_scalars, size_t n_scalars, int require_equal_len,
                                uint64_t* h_out_xy, uint8_t* h_out_inf) {
    if (!c || !h_out_xy || !h_out_inf) return ZKHIP_ERR_ARG;
    if (require_equal_len && n_points != n_scalars) return ZKHIP_ERR_SHAPE;   // multilinear_kzg.rs:36-41
    if (n_scalars > n_points) return ZKHIP_ERR_INDEX;                          // univariate_kzg.rs:53
    const size_t n = n_scalars;
    if (n == 0) { std::memset(h_out_xy, 0, 96); *h_out_inf = 1; return ZKHIP_OK; }   // P::G1::default()
    if (!d_points_xy || !d_scalars) return ZKHIP_ERR_ARG;
    if (n >= ((size_t)1 << 31)) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    const MsmPlan pl = msm_plan(n);
    const size_t n_buckets = (size_t)pl.n_windows * pl.nb;
    const size_t n_segments = (size_t)pl.n_windows * pl.ns;
    const size_t n_out = (size_t)pl.n_windows * pl.n_terms;
    // workspace carve-up (all 256-byte aligned)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_counts = 0;
    const size_t o_offsets = o_counts + al(n_buckets * 4);
    const size_t o_cursor = o_offsets + al(n_buckets * 4);
    const size_t o_sorted = o_cursor + al(n_buckets * 4);
    const size_t o_buckets = o_sorted + al(n * pl.n_windows * 4);
    const size_t o_segs = o_buckets + al(n_buckets * 192);
    const size_t o_sega = o_segs + al(n_segments * 192);
    const size_t o_terms = o_sega + al(n_segments * 192);
    const size_t total = o_terms + al(n_out * 192);
    ZK_TRY(c->reserve_ws(total));
    char* ws = (char*)c->d_ws;
    uint32_t* counts = (uint32_t*)(ws + o_counts);
    uint32_t* offsets = (uint32_t*)(ws + o_offsets);
    uint32_t* cursor = (uint32_t*)(ws + o_cursor);
    uint32_t* sorted = (uint32_t*)(ws + o_sorted);
    uint64_t* buckets = (uint64_t*)(ws + o_buckets);
    uint64_t* segs = (uint64_t*)(ws + o_segs);
    uint64_t* sega = (uint64_t*)(ws + o_sega);
    uint64_t* terms = (uint64_t*)(ws + o_terms);

    ZK_HIP(c, hipMemsetAsync(counts, 0, n_buckets * 4, c->stream));
    const int grid_n = (int)std::min<size_t>((n + MSM_BLOCK - 1) / MSM_BLOCK, 256 * 8);
    {
        ProfScope ps(c, "msm_hist", 32.0 * (double)n);
        hipLaunchKernelGGL(msm_hist_kernel, dim3(grid_n), dim3(MSM_BLOCK), 0, c->stream, d_scalars, d_points_inf, n, pl, counts);
    }
    hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(1024), 0, c->stream, counts, (uint32_t)n_buckets, offsets, cursor);
    {
        ProfScope ps(c, "msm_scatter", 32.0 * (double)n);
        hipLaunchKernelGGL(msm_scatter_kernel, dim3(grid_n), dim3(MSM_BLOCK), 0, c->stream, d_scalars, d_points_inf, n, pl, cursor, sorted);
    }
    {
        ProfScope ps(c, "msm_accumulate", 128.0 * (double)n);
        hipLaunchKernelGGL(msm_accumulate_kernel, dim3((unsigned)((n_buckets + MSM_BLOCK - 1) / MSM_BLOCK)), dim3(MSM_BLOCK), 0,
                           c->stream, d_points_xy, sorted, offsets, counts, (uint32_t)n_buckets, buckets);
    }
    {
        ProfScope ps(c, "msm_segment", 0.0);
        hipLaunchKernelGGL(msm_segment_kernel, dim3((unsigned)((n_segments + MSM_BLOCK - 1) / MSM_BLOCK)), dim3(MSM_BLOCK), 0,
                           c->stream, buckets, (uint32_t)n_segments, segs, sega);
    }
    {
        ProfScope ps(c, "msm_terms", 0.0);
        hipLaunchKernelGGL(msm_terms_kernel, dim3((unsigned)n_out), dim3(MSM_BLOCK), MSM_BLOCK * 192, c->stream, segs, sega, pl, terms);
    }
    ZK_HIP(c, hipGetLastError());
    std::vector<uint64_t> h_terms(n_out * 24);
    ZK_HIP(c, hipMemcpyAsync(h_terms.data(), terms, n_out * 192, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    // host epilogue: sum over (window w, term t) of 2^exp * point
    std::vector<zkhost::Xyzz> pts(n_out);
    std::vector<uint32_t> exps(n_out);
    for (size_t i = 0; i < n_out; ++i) {
        std::memcpy(&pts[i], &h_terms[24 * i], 192);
        const uint32_t w = (uint32_t)(i / pl.n_terms), t = (uint32_t)(i % pl.n_terms);
        exps[i] = w * pl.c + (t == 0 ? 0 : MSM_SEG_LOG + (t - 1));
    }
    zkhost::Xyzz res = zkhost::weighted_sum_pow2(pts, exps);
    *h_out_inf = zkhost::xyzz_to_affine(res, h_out_xy) ? 0 : 1;
    return ZKHIP_OK;
}

// scalars (device, n x 4) -> affine SRS points
static int srs_from_scalars(zkhip_ctx* c, const uint64_t* d_scalars, size_t n, uint64_t* d_out_xy, uint8_t* d_out_inf) {
    // workspace layout: [scalars n*32 (owned by caller region)] ... we only need n*192 for XYZZ here
    uint64_t* xyzz = (uint64_t*)((char*)c->d_ws + ((n * 32 + 255) & ~(size_t)255));
    hipLaunchKernelGGL(srs_fixed_base_kernel, dim3((unsigned)((n + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0, c->stream,
                       d_scalars, n, xyzz);
    const size_t n_threads = (n + SRS_CHUNK - 1) / SRS_CHUNK;
    hipLaunchKernelGGL(srs_batch_affine_kernel, dim3((unsigned)((n_threads + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0,
                       c->stream, xyzz, n, d_out_xy, d_out_inf);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

extern "C" int zkhip_srs_multilinear_g1(zkhip_ctx* c, const uint64_t* h_tau, uint32_t n_vars, uint64_t* d_out_xy,
                                        uint8_t* d_out_inf) {
    if (!c || !d_out_xy || !d_out_inf || (n_vars && !h_tau)) return ZKHIP_ERR_ARG;
    if (n_vars > 30 || n_vars > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    const size_t n = (size_t)1 << n_vars;
    ZK_TRY(c->reserve_ws(((n * 32 + 255) & ~(size_t)255) + n * 192));
    uint64_t* d_tau = c->small_u64(ZK_SMALL_PTS);
    if (n_vars) {
        std::memcpy(c->pinned_u64(ZK_PIN_PTS), h_tau, 32 * (size_t)n_vars);
        ZK_HIP(c, hipMemcpyAsync(d_tau, c->pinned_u64(ZK_PIN_PTS), 32 * (size_t)n_vars, hipMemcpyHostToDevice, c->stream));
    }
    uint64_t* d_scalars = (uint64_t*)c->d_ws;
    hipLaunchKernelGGL(srs_eq_scalars_kernel, dim3(mle_grid(n)), dim3(SRS_BLOCK), 0, c->stream, d_tau, n_vars, d_scalars);
    return srs_from_scalars(c, d_scalars, n, d_out_xy, d_out_inf);
}

extern "C" int zkhip_srs_univariate_g1(zkhip_ctx* c, const uint64_t* h_tau, size_t max_degree, uint64_t* d_out_xy,
                                       uint8_t* d_out_inf) {
    if (!c || !d_out_xy || !d_out_inf || !h_tau) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    const size_t n = max_degree + 1;
    ZK_TRY(c->reserve_ws(((n * 32 + 255) & ~(size_t)255) + n * 192));
    uint64_t* d_tau = c->small_u64(ZK_SMALL_R);
    std::memcpy(c->pinned_u64(ZK_PIN_R), h_tau, 32);
    ZK_HIP(c, hipMemcpyAsync(d_tau, c->pinned_u64(ZK_PIN_R), 32, hipMemcpyHostToDevice, c->stream));
    uint64_t* d_scalars = (uint64_t*)c->d_ws;
    hipLaunchKernelGGL(srs_power_scalars_kernel, dim3(mle_grid(n)), dim3(SRS_BLOCK), 0, c->stream, d_tau, n, d_scalars);
    return srs_from_scalars(c, d_scalars, n, d_out_xy, d_out_inf);
}

extern "C" int zkhip_g1_sum_affine(const uint64_t* h_points_xy, const uint8_t* h_points_inf, size_t n,
                                   uint64_t* h_out_xy, uint8_t* h_out_inf) {
    if ((n && !h_points_xy) || !h_out_xy || !h_out_inf) return ZKHIP_ERR_ARG;
    zkhost::Xyzz acc = zkhost::xyzz_identity();
    for (size_t i = 0; i < n; ++i)
        acc = zkhost::xyzz_add(acc, zkhost::xyzz_from_affine(h_points_xy + 12 * i, h_points_inf && h_points_inf[i]));
    *h_out_inf = zkhost::xyzz_to_affine(acc, h_out_xy) ? 0 : 1;
    return ZKHIP_OK;
}

#ifdef ZK_STAMPS
extern "C" int zkhip_debug_read_stamps(zkhip_ctx* c, unsigned long long* h_out /*64*8*/) {
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    ZK_HIP(c, hipMemcpyFromSymbol(h_out, HIP_SYMBOL(zk::g_zk_stamps), 64 * 8 * 8));
    return ZKHIP_OK;
}
#endif

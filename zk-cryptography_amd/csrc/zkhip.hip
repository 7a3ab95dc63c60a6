// zkhip.hip -- C-ABI entry points (include/zkhip.h) and host-side launch logic.
// gfx950 only.  No CPU fallback: every entry point launches HIP kernels or fails.
#include "../../include/zkhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "host_util.hpp"
#include "mle_kernels.hpp"
#include "sumcheck_kernels.hpp"
#include "multifold_kernels.hpp"
#include "mfma_fold.hpp"

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
        case ZKHIP_ERR_BUSY: return "workspace lent to a live split-phase session";
        case ZKHIP_ERR_PEER: return "another rank of the sharded prover failed";
        case ZKHIP_ERR_TIMEOUT: return "a device-side wait gave up";
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
    for (zkhip_ctx* lane : c->gkr_lanes) zkhip_ctx_destroy(lane);      // zkhip_gkr_prove_batch's lanes
    c->gkr_lanes.clear();
    hipSetDevice(c->device);
    for (int k = 0; k < zkhip_ctx::ASYNC_SLOTS; ++k) if (c->async_pend[k]) zkhip_kzg_commit_end(c, (uint32_t)k, nullptr, nullptr);   // commits never collected
    (void)c->flush_deferred();
    hipStreamSynchronize(c->stream);
    if (c->gkr_graph.exec) (void)hipGraphExecDestroy((hipGraphExec_t)c->gkr_graph.exec);
    if (c->d_gkr_in) (void)hipFree(c->d_gkr_in);
    // proofs still in flight write their results into the pinned slots from their lanes' streams: drain every stream of the context first
    if (c->fold_stream) hipStreamSynchronize(c->fold_stream);
    for (auto& L : c->lanes) { if (L.serial) hipStreamSynchronize(L.serial); if (L.fold) hipStreamSynchronize(L.fold); }
    for (int i = 0; i < zkhip_ctx::MSM_SLOTS; ++i) if (c->side[i]) hipStreamSynchronize(c->side[i]);
    for (auto& e : c->prof_events) { hipEventDestroy(e.start); hipEventDestroy(e.stop); }
    if (c->d_ws) hipFree(c->d_ws);
    if (c->d_aux) hipFree(c->d_aux);
    if (c->d_gen_table) hipFree(c->d_gen_table);
    if (c->d_composed) hipFree(c->d_composed);
    if (c->d_fingerprint) hipFree(c->d_fingerprint);
    if (c->guard_stream) { hipStreamSynchronize(c->guard_stream); hipStreamDestroy(c->guard_stream); }
    if (c->guard_ev) hipEventDestroy(c->guard_ev);
    if (c->ntt_state && c->ntt_free) c->ntt_free(c->ntt_state);
    for (int i = 0; i < zkhip_ctx::MSM_SLOTS; ++i) {
        if (c->msm_pin[i]) hipHostFree(c->msm_pin[i]);
        if (c->msm_tab_dev[i]) hipFree(c->msm_tab_dev[i]);
        if (c->msm_tab_pin[i]) hipHostFree(c->msm_tab_pin[i]);
        if (c->msm_ev[i]) hipEventDestroy(c->msm_ev[i]);
        if (c->side[i]) hipStreamDestroy(c->side[i]);
    }
    delete c->host_pool;
    if (c->fork_ev) hipEventDestroy(c->fork_ev);
    if (c->join_ev) hipEventDestroy(c->join_ev);
    if (c->serial_ev) hipEventDestroy(c->serial_ev);
    if (c->done_ev) hipEventDestroy(c->done_ev);
    for (int k = 0; k < zkhip_ctx::PROOF_SLOTS; ++k) { if (c->proof_ev[k]) hipEventDestroy(c->proof_ev[k]); if (c->proof_pin[k]) hipHostFree(c->proof_pin[k]); }
    if (c->fold_stream) hipStreamDestroy(c->fold_stream);
    for (int k = 0; k < zkhip_ctx::COARSE_RING; ++k) if (c->d_coarse[k]) hipFree(c->d_coarse[k]);
    for (auto& L : c->lanes) {
        if (L.serial && !L.borrowed) hipStreamDestroy(L.serial);
        if (L.fold && !L.borrowed) hipStreamDestroy(L.fold);
        if (L.begin_ev) hipEventDestroy(L.begin_ev);
        if (L.fork_ev) hipEventDestroy(L.fork_ev);
        if (L.serial_ev) hipEventDestroy(L.serial_ev);
        if (L.ws) hipFree(L.ws);
        if (L.small) hipFree(L.small);
    }
    if (c->d_small) hipFree(c->d_small);
    if (c->sc_small) hipFree(c->sc_small);
    if (c->sc_stage) hipFree(c->sc_stage);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return ZKHIP_OK;
}

extern "C" int zkhip_ctx_set_stream(zkhip_ctx* c, void* stream) {
    if (!c) return ZKHIP_ERR_ARG;
    if ((hipStream_t)stream == c->stream) return ZKHIP_OK;
    // Several entry points return before their kernels have run, and they share the context's scratch buffers: work
    // enqueued on the new stream must come after what is still queued on the old one.
    ZK_TRY(c->activate());
    (void)c->flush_deferred();                  // proofs in flight finish on the stream they began on (a failure is reported by their prove_end)
    ZK_TRY(c->ensure_side_streams());
    ZK_HIP(c, hipEventRecord(c->fork_ev, c->stream));
    ZK_HIP(c, hipStreamWaitEvent((hipStream_t)stream, c->fork_ev, 0));
    c->stream = (hipStream_t)stream;
    return ZKHIP_OK;
}
extern "C" int zkhip_ctx_synchronize(zkhip_ctx* c) {
    if (!c) return ZKHIP_ERR_ARG;
    (void)c->flush_deferred();
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
    if (d_ptr) (void)zkhip_table_release(c, d_ptr);      // (a table of the commit path: its address is no longer a known table, msm.hip)
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

extern "C" int zkhip_profile_timeline(zkhip_ctx* c, uint32_t max_records, char* names, double* start_us, double* stop_us,
                                      uint32_t* count) {
    if (!c || !names || !start_us || !stop_us || !count) return ZKHIP_ERR_ARG;
    ZK_HIP(c, hipDeviceSynchronize());
    uint32_t n = 0;
    for (auto& r : c->prof_records) {
        if (n == max_records) break;
        float a = 0, b = 0;
        ZK_HIP(c, hipEventElapsedTime(&a, c->prof_events[c->prof_records[0].event].start, c->prof_events[r.event].start));
        ZK_HIP(c, hipEventElapsedTime(&b, c->prof_events[c->prof_records[0].event].start, c->prof_events[r.event].stop));
        std::snprintf(names + 32 * (size_t)n, 32, "%s", r.name);
        start_us[n] = 1e3 * a;
        stop_us[n] = 1e3 * b;
        ++n;
    }
    *count = n;
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------

// fold variable var_index of an n-entry table; r on device. with_sums -> partials (returns grid size in *np)
static int launch_fold(zkhip_ctx* c, const uint64_t* d_in, size_t n, const uint64_t* d_r, const uint64_t* h_r,
                       uint32_t var_index, uint64_t* d_out, bool with_sums, uint64_t* d_partials, uint32_t* np) {
    FrArg rv = {};
    if (h_r) std::memcpy(rv.v, h_r, 32);
    const size_t n_out = n / 2;
    const uint32_t log_half = log2_exact(n) - 1 - var_index;
    const int grid = with_sums ? mle_grid((n_out + 1) / 2) : mle_grid_stream((n_out + 1) / 2);   // with sums: one record per workgroup
    ProfScope ps(c, with_sums ? "fold_sums" : "fold", 48.0 * (double)n);
    if (with_sums)
        hipLaunchKernelGGL(fold_kernel<true>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_in, d_out, n_out, log_half,
                           d_r, rv, d_partials);
    else {
        // at most THREE workgroups per CU (a 40 KiB LDS request the kernel never touches; section 7 of profiles/r06/NOTES.md found the sums pass
        // no slower with a quarter of the loads in flight): 0.1427 -> 0.1372 ms at 2^24, 0.70 -> 0.73 of HBM; two per CU 0.161, four 0.137, no cap 0.143.
        // ZKHIP_FOLD_LDS=<bytes> overrides (diagnostics; 0 = no cap).
        static const long fold_lds = [] { const char* e = getenv("ZKHIP_FOLD_LDS"); return e ? std::min(std::max(atol(e), 0L), 158L * 1024) : 40960L; }();
        if (fold_lds > 64 * 1024) ZK_TRY(c->allow_big_lds((const void*)fold_kernel<false>, 158 * 1024));
        hipLaunchKernelGGL(fold_kernel<false>, dim3(grid), dim3(MLE_BLOCK), (size_t)fold_lds, c->stream, d_in, d_out, n_out, log_half,
                           d_r, rv, d_partials);
    }
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
    if (var_index >= log2_exact(n)) return ZKHIP_OK;                   // 2^k >= n: the pair list is empty (utils.rs:37-50), so is the table
    ZK_TRY(c->activate());
    return launch_fold(c, d_evals, n, d_r, h_r, var_index, d_out, false, nullptr, nullptr);
}

static int launch_multifold(zkhip_ctx* c, hipStream_t stream, const uint64_t* cur, size_t cn, uint32_t k, const uint64_t* d_w,
                            uint64_t* dst, uint64_t* pdst, uint32_t* n_parts);
// Successive folds of variable 0 (or of h_var_indices); the points are host values, passed by value per launch.
static int fold_chain(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_pts,
                      const uint32_t* var_indices, size_t n_pts, uint64_t* d_out) {
    PtsArg pa = {};
    if (n_pts) std::memcpy(pa.v, h_pts, 32 * n_pts);
    // ping-pong buffers: A holds n/2, B holds n/4
    if (n_pts == 0) {
        ZK_HIP(c, hipMemcpyAsync(d_out, d_evals, n * 32, hipMemcpyDeviceToDevice, c->stream));
        return ZKHIP_OK;
    }
    const size_t need = (n / 2 + n / 4 + 8) * 32 + (256 + n / 512 + 4096) * 32;
    ZK_TRY(c->reserve_ws(need));
    uint64_t* A = (uint64_t*)c->d_ws;
    uint64_t* B = A + 4 * (n / 2);
    uint64_t* d_w = B + 4 * (n / 4 + 4);
    uint64_t* d_dummy = d_w + 4 * 256;            // per-workgroup output sums of the k-variable fold (unused here; <= n / 512 of them)
    const uint64_t* cur = d_evals;
    size_t cn = n;
    size_t p0 = 0;
    // every point folds variable 0 (`evaluation`, and partial_evaluations(&r, &vec![0; k]) as GKR calls it, gkr/src/protocol.rs:64-68):
    // collapse k of them per pass (weights = eq table of those points)
    bool all_zero = true;
    if (var_indices) for (size_t q = 0; q < n_pts; ++q) all_zero = all_zero && var_indices[q] == 0;
    if (all_zero) {
        uint32_t stage = 0;
        while (cn > (size_t)TAIL_N && n_pts - p0 >= 3) {
            uint32_t k = log2_exact(cn) - TAIL_LOG;
            if (k > MF_MAX_LOGK) k = MF_MAX_LOGK;
            if (k > n_pts - p0) k = (uint32_t)(n_pts - p0);
            if (k < 3) break;
            const size_t m = cn >> k;
            const bool last = (p0 + k == n_pts);
            uint64_t* dst = last ? d_out : ((stage & 1) ? B : A);
            hipLaunchKernelGGL(eq_weights_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, pa, (uint32_t)p0, k, d_w);
            ZK_TRY(launch_multifold(c, c->stream, cur, cn, k, d_w, dst, d_dummy, nullptr));
            cur = dst; cn = m; p0 += k; ++stage;
            if (last) return ZKHIP_OK;
        }
    }
    for (size_t p = p0; p < n_pts; ++p) {
        const uint32_t k = var_indices ? var_indices[p] : 0;
        if (cn < 2 || !((size_t)k < cn / 2)) return ZKHIP_ERR_SHAPE;
        if (k >= log2_exact(cn)) return p + 1 == n_pts ? ZKHIP_OK : ZKHIP_ERR_SHAPE;   // empty table (utils.rs:37-50); a further fold of it panics
        const bool all_zero_tail = !var_indices || [&] {
            for (size_t q = p; q < n_pts; ++q) if (var_indices[q]) return false;
            return true;
        }();
        if (all_zero_tail && cn <= (size_t)TAIL_N) {
            // finish every remaining fold inside one workgroup
            ProfScope ps(c, "fold_tail", 0.0);
            ZK_TRY(c->allow_big_lds((const void*)fold_tail_kernel, TAIL_LDS_BYTES));
            hipLaunchKernelGGL(fold_tail_kernel, dim3(1), dim3(MLE_BLOCK), TAIL_LDS_BYTES, c->stream, cur, (uint32_t)cn,
                               pa, (uint32_t)p, (uint32_t)(n_pts - p), d_out);
            ZK_HIP(c, hipGetLastError());
            return ZKHIP_OK;
        }
        const bool last = (p + 1 == n_pts);
        uint64_t* dst = last ? d_out : ((cur == A) ? B : A);
        ZK_TRY(launch_fold(c, cur, cn, nullptr, h_pts + 4 * p, k, dst, false, nullptr, nullptr));
        cur = dst;
        cn /= 2;
    }
    return ZKHIP_OK;
}

extern "C" size_t zkhip_mle_partial_evaluation_len(size_t n, uint32_t var_index) {
    if (!is_pow2(n) || n < 2 || !((size_t)var_index < n / 2)) return 0;
    return var_index >= log2_exact(n) ? 0 : n / 2;
}

extern "C" int zkhip_mle_partial_evaluations(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_pts,
                                             const uint32_t* h_var_indices, size_t n_pts, uint64_t* d_out) {
    if (!c || !d_evals || !d_out || (n_pts && (!h_pts || !h_var_indices))) return ZKHIP_ERR_ARG;
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;
    if (n_pts > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    return fold_chain(c, d_evals, n, h_pts, h_var_indices, n_pts, d_out);
}

extern "C" int zkhip_mle_evaluation(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_pts,
                                    size_t n_pts, uint64_t* h_out) {
    if (!c || !d_evals || !h_out || (n_pts && !h_pts)) return ZKHIP_ERR_ARG;
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;
    if (n_pts != log2_exact(n)) return ZKHIP_ERR_SHAPE;                // assert_eq! evaluation_form.rs:163-167
    ZK_TRY(c->activate());
    uint64_t* d_res = c->small_u64(ZK_SMALL_RES);
    // ZKHIP_EVAL_ONE_PASS=0 (diagnostics): the chain of k-variable folds that partial_evaluations takes
    static const bool one_pass = [] { const char* e = getenv("ZKHIP_EVAL_ONE_PASS"); return !e || atoi(e) != 0; }();
    const uint32_t log_n = log2_exact(n);
    if (one_pass && log_n >= 17) {
        // Every point is known before the first launch, so the evaluation is ONE pass over the table: with k1 leading variables per
        // output, p(r) = sum_j eq_j(r[k1..]) * (sum_b eq_b(r[..k1]) * T[b * m + j]) -- the k1-variable fold on the matrix cores
        // whose tiles weight their 64 outputs with the eq table of the remaining points and keep only that sum (32 n bytes read,
        // nothing written but m / 64 records), then the records' sum.  Three launches: the weight tables, the pass, the sum.
        const uint32_t k1 = std::min<uint32_t>(MF_MAX_LOGK, log_n - 13), k2 = log_n - k1, s2 = k2 / 2;
        const size_t m = n >> k1, tiles = m / 64, na = (size_t)1 << (k2 - s2), nb = (size_t)1 << s2;
        ZK_TRY(c->reserve_ws((256 + na + nb + tiles + 16) * 32));
        uint64_t* d_w1 = (uint64_t*)c->d_ws;
        uint64_t* d_wa = d_w1 + 4 * 256;
        uint64_t* d_wb = d_wa + 4 * na;
        uint64_t* d_rec = d_wb + 4 * nb;
        PtsArg pa = {};
        std::memcpy(pa.v, h_pts, 32 * n_pts);
        const size_t lanes = ((size_t)1 << k1) + na + nb;
        hipLaunchKernelGGL(eval_weights_kernel, dim3((unsigned)((lanes + MLE_BLOCK - 1) / MLE_BLOCK)), dim3(MLE_BLOCK), 0, c->stream, pa, k1, k2, s2, d_w1, d_wa, d_wb);
        {
            ProfScope ps(c, "multifold_eval", 32.0 * (double)n);
            const size_t q_bytes = mfm_lds_bytes(std::min<uint32_t>(1u << k1, (uint32_t)MFM_CHUNK));
            if (q_bytes > 64 * 1024) ZK_TRY(c->allow_big_lds((const void*)multifold_mfma_kernel<4, 4, true>, 158 * 1024));
            hipLaunchKernelGGL((multifold_mfma_kernel<4, 4, true>), dim3((unsigned)(tiles / 4)), dim3(256), q_bytes, c->stream, d_evals, m, k1,
                               (const uint64_t*)d_w1, (uint64_t*)nullptr, d_rec, 1u, (const uint64_t*)d_wa, (const uint64_t*)d_wb, s2);
        }
        hipLaunchKernelGGL(sum_records_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, (const uint64_t*)d_rec, (uint32_t)tiles, d_res);
        ZK_HIP(c, hipGetLastError());
    } else {
        ZK_TRY(fold_chain(c, d_evals, n, h_pts, nullptr, n_pts, d_res));
    }
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
    const int grid = mle_grid_stream(n_out);
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
                                     const uint64_t* h_scalar, size_t n, size_t nb, uint64_t* d_out) {
    if (!c || !d_a || !d_out || op < 0 || op > 2) return ZKHIP_ERR_ARG;
    if (op == 2 ? !h_scalar : !d_b) return ZKHIP_ERR_ARG;
    if (op != 2 && nb < n) return ZKHIP_ERR_INDEX;      // rhs.evaluations[i] out of bounds (evaluation_form.rs:185,215)
    if (n == 0) return ZKHIP_OK;
    ZK_TRY(c->activate());
    const int grid = mle_grid_stream(n);
    FrArg sc = {};
    if (op == 2) {
        std::memcpy(sc.v, h_scalar, 32);
        hipLaunchKernelGGL(elementwise_kernel<2>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_a, (const uint64_t*)nullptr, sc, n, d_out);
    } else if (op == 0) {
        hipLaunchKernelGGL(elementwise_kernel<0>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_a, d_b, sc, n, d_out);
    } else {
        hipLaunchKernelGGL(elementwise_kernel<1>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_a, d_b, sc, n, d_out);
    }
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

extern "C" int zkhip_mle_to_bytes(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint8_t* d_out_bytes) {
    if (!c || !d_evals || !d_out_bytes) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    hipLaunchKernelGGL(to_bytes_kernel, dim3(mle_grid_stream(n)), dim3(MLE_BLOCK), 0, c->stream, d_evals, n,
                       (uint32_t*)d_out_bytes);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

extern "C" int zkhip_mle_add_to_front(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint32_t variable_length, uint64_t* d_out) {
    if (!c || !d_evals || !d_out) return ZKHIP_ERR_ARG;
    if (!is_pow2(n) || variable_length > 40) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    const size_t n_out = n * ((size_t)2 << variable_length);
    hipLaunchKernelGGL(repeat_kernel, dim3(mle_grid_stream(n_out)), dim3(MLE_BLOCK), 0, c->stream, d_evals, n, 0u, n_out, d_out);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}
extern "C" int zkhip_mle_add_to_back(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint32_t variable_length, uint64_t* d_out) {
    if (!c || !d_evals || !d_out) return ZKHIP_ERR_ARG;
    if (!is_pow2(n) || variable_length > 40) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    if (variable_length == 0) {
        ZK_HIP(c, hipMemcpyAsync(d_out, d_evals, 32 * n, hipMemcpyDeviceToDevice, c->stream));
        return ZKHIP_OK;
    }
    const size_t n_out = n << variable_length;
    hipLaunchKernelGGL(repeat_kernel, dim3(mle_grid_stream(n_out)), dim3(MLE_BLOCK), 0, c->stream, d_evals, n, variable_length, n_out, d_out);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// layered circuit (GKR table builders)
// ---------------------------------------------------------------------------------------
extern "C" size_t zkhip_gkr_mle_size(uint32_t layer_index) {   // circuit/src/utils.rs:1-10
    if (layer_index == 0) return (size_t)1 << 3;
    return (size_t)1 << (layer_index + 2 * (layer_index + 1));
}
// packs the host gate arrays into the workspace as (type, in0, in1) words
static int upload_gates(zkhip_ctx* c, const uint8_t* h_gate_type, const uint32_t* h_in0, const uint32_t* h_in1, size_t n_gates,
                        uint32_t** d_gates) {
    std::vector<uint32_t> packed(3 * n_gates);
    for (size_t g = 0; g < n_gates; ++g) {
        packed[3 * g] = h_gate_type[g] ? 1u : 0u;
        packed[3 * g + 1] = h_in0[g];
        packed[3 * g + 2] = h_in1[g];
    }
    ZK_TRY(c->reserve_ws(12 * n_gates + 256));
    ZK_HIP(c, hipMemcpyAsync(c->d_ws, packed.data(), 12 * n_gates, hipMemcpyHostToDevice, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));   // `packed` is a stack temporary
    *d_gates = (uint32_t*)c->d_ws;
    return ZKHIP_OK;
}
extern "C" int zkhip_circuit_layer_eval(zkhip_ctx* c, const uint64_t* d_in, size_t n_in, const uint8_t* h_gate_type,
                                        const uint32_t* h_in0, const uint32_t* h_in1, size_t n_gates, uint64_t* d_out) {
    if (!c || !d_in || !d_out || (n_gates && (!h_gate_type || !h_in0 || !h_in1))) return ZKHIP_ERR_ARG;
    for (size_t g = 0; g < n_gates; ++g)
        if (h_in0[g] >= n_in || h_in1[g] >= n_in) return ZKHIP_ERR_INDEX;   // current_input[e.inputs[k]] panics
    if (n_gates == 0) return ZKHIP_OK;
    ZK_TRY(c->activate());
    uint32_t* d_gates = nullptr;
    ZK_TRY(upload_gates(c, h_gate_type, h_in0, h_in1, n_gates, &d_gates));
    hipLaunchKernelGGL(circuit_layer_kernel, dim3(mle_grid(n_gates)), dim3(MLE_BLOCK), 0, c->stream, d_in, d_gates, n_gates, d_out);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipStreamSynchronize(c->stream));   // the gate list lives in the shared workspace
    return ZKHIP_OK;
}
extern "C" int zkhip_circuit_add_mult_mle(zkhip_ctx* c, const uint8_t* h_gate_type, const uint32_t* h_in0, const uint32_t* h_in1,
                                          size_t n_gates, uint32_t layer_index, uint64_t* d_add, uint64_t* d_mul) {
    if (!c || !d_add || !d_mul || (n_gates && (!h_gate_type || !h_in0 || !h_in1))) return ZKHIP_ERR_ARG;
    if (layer_index > 12) return ZKHIP_ERR_SHAPE;   // 2^(3 l + 2) elements
    const size_t size = zkhip_gkr_mle_size(layer_index);
    const uint32_t shift = layer_index + 1;
    for (size_t g = 0; g < n_gates; ++g) {
        const size_t idx = (g << (2 * shift)) | ((size_t)h_in0[g] << shift) | h_in1[g];
        if (idx >= size) return ZKHIP_ERR_INDEX;     // add_evaluations[gate_decimal] out of bounds
    }
    ZK_TRY(c->activate());
    ZK_HIP(c, hipMemsetAsync(d_add, 0, 32 * size, c->stream));   // F::zero() is all-zero limbs
    ZK_HIP(c, hipMemsetAsync(d_mul, 0, 32 * size, c->stream));
    if (n_gates == 0) return ZKHIP_OK;
    uint32_t* d_gates = nullptr;
    ZK_TRY(upload_gates(c, h_gate_type, h_in0, h_in1, n_gates, &d_gates));
    hipLaunchKernelGGL(wiring_ones_kernel, dim3(mle_grid(n_gates)), dim3(MLE_BLOCK), 0, c->stream, d_gates, n_gates, shift, d_add, d_mul);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// basic sumcheck prover
// ---------------------------------------------------------------------------------------
// Stage plan of the prover: a table of cur_n > TAIL_N entries is folded by k variables at once, k chosen so
// that tables up to 2^20 entries land exactly on the LDS tail and larger ones shrink 256-fold per pass.
static inline uint32_t stage_k(size_t cur_n) {
    if (cur_n <= ((size_t)1 << TREE_MAX_LOG)) return 0;      // finishes inside the serial kernel
    const uint32_t lg = log2_exact(cur_n);
    uint32_t k = lg - 8;                                       // aim at a 2^8-entry table (1 fold product per helper lane)
    if (k > MF_MAX_LOGK) k = MF_MAX_LOGK;
    return k;                                                  // >= 3 here
}
// Overlapped plan (tables of 2^24 entries; measured against the stage plan at every size with the rounds at 4.5 us, tools/step_sizes.py:
// 2^19 192 / 158 us, 2^20 201 / 171, 2^21 216 / 199, 2^22 230 / 218, 2^23 259 / 253, 2^24 322 / 347 -- only where the fold is long
// enough to hide ten rounds does running them beside it pay for the two extra launches).  Identity (1) of multifold_kernels.hpp applied twice: with FINE block
// sums B_g (2^g blocks of 256 entries, g = lg - 8) the first k1 rounds run on B_k1 (grouped B_g) and the next k2 = g - k1
// rounds on fold_k1(B_g; r_1..r_k1) -- a table of 2^k2 <= 1024 entries -- so rounds k1+1 .. g need the first k1 challenges
// but NOT the folded big table.  The streaming k1-variable fold of the big table therefore runs on a side stream next to
// the serial kernel of those rounds and is joined before the fold by the next k2 variables.  k2 is as large as the serial
// kernel takes (10), so that the fold starts as early as possible: 2^24 = 6 rounds | fold (100 us) next to 10 rounds | 8
// rounds.  Same values as the round-by-round loop, bit for bit.
static inline uint32_t overlapped_min_log() {       // (ZKHIP_OVERLAP_MIN_LOG: tuning aid, read once)
    static const uint32_t v = [] { const char* e = std::getenv("ZKHIP_OVERLAP_MIN_LOG"); const int x = e ? std::atoi(e) : 0; return x >= 19 && x <= 25 ? (uint32_t)x : 24u; }();
    return v;
}
static inline bool overlapped_plan(size_t n) { return is_pow2(n) && n >= ((size_t)1 << overlapped_min_log()) && n <= ((size_t)1 << 24); }
// A rank's SHARD takes the overlapped stage over the whole range it was built for: there it is also the form with the fewest exchanges
// (three per proof), which is what counts over the fabric.
static inline bool overlapped_shard(size_t n) { return is_pow2(n) && n >= ((size_t)1 << 19) && n <= ((size_t)1 << 24); }
static inline uint32_t overlapped_k2(size_t n) { return std::min<uint32_t>(TREE_MAX_LOG, log2_exact(n) - 8 - 3); }
extern "C" int zkhip_sumcheck_plan_log_blocks(size_t n) {
    if (!is_pow2(n)) return 0;
    return overlapped_plan(n) ? (int)(log2_exact(n) - 8) : (int)stage_k(n);
}

// min_lds: pads the dynamic LDS request.  The serial kernel that runs next to the streaming fold asks for (nearly) a whole
// CU's LDS so that no workgroup of the fold shares its CU: beside 8+ fold waves per SIMD the transcript wave ran at half speed.
static int launch_small(zkhip_ctx* c, const SmallArgs& a, SumcheckDev* st, uint64_t* d_rp, uint64_t* d_ch, size_t min_lds = 0,
                        hipStream_t stream = nullptr) {
    if (!stream) stream = c->stream;
    if (a.log_n > (uint32_t)TREE_MAX_LOG || a.n_rounds > a.log_n) return ZKHIP_ERR_SHAPE;
    const size_t lds = std::max(small_lds_bytes(a.log_n, a.weights_out ? (int)a.n_rounds : -1, a.group != 0 && a.stride == 0), min_lds);
    if (lds > 160 * 1024) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->allow_big_lds((const void*)sumcheck_small_kernel, 160 * 1024));
    ProfScope ps(c, "sumcheck_small", 0.0, stream);
    hipLaunchKernelGGL(sumcheck_small_kernel, dim3(1), dim3(SMALL_BLOCK), lds, stream, a, st, d_rp, d_ch);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

// the k-variable fold of a cn-entry table in the shape that suits its output size; returns the per-workgroup sums' count
static int launch_multifold(zkhip_ctx* c, hipStream_t stream, const uint64_t* cur, size_t cn, uint32_t k, const uint64_t* d_w,
                            uint64_t* dst, uint64_t* pdst, uint32_t* n_parts) {
    const size_t m = cn >> k;
    uint32_t out_per_wg;
    // diagnostics: ZKHIP_MF=0 keeps the VALU form for an A/B on the same box (tools/ab_multifold.sh); ZKHIP_MF=r (1..9) sets the rotation of
    // the term order (tile T starts at term r * T mod 2^k; the default 1 -- 3 and 5 measured the same, 0 = every wave at term 0: 108 vs 112 us
    // before the aligned planes); ZKHIP_MF_OCC=n caps the workgroups per CU through the LDS request (all 1024 workgroups are resident at once as it is, 4 per CU; capped at 2 per CU
    // or with 36-78 KiB requested per workgroup the pass takes the same 83-85 us, alone and beside the proofs in flight: round 6)
    static const int mf_cfg = [] { const char* e = getenv("ZKHIP_MF"); return e ? atoi(e) : 1; }();
    if (m >= 8192 && k >= 4 && mf_cfg != 0) {   // streaming shape, limb products on the matrix cores (mfma_fold.hpp)
        out_per_wg = 64;
        ProfScope ps(c, "multifold", 32.0 * (double)cn + 32.0 * (double)m, stream);
        const unsigned tiles = (unsigned)(m / 64), rot = (unsigned)(mf_cfg % 10);
        static const int mf_occ = [] { const char* e = getenv("ZKHIP_MF_OCC"); return e ? atoi(e) : 0; }();
        size_t q_bytes = mfm_lds_bytes(std::min<uint32_t>(1u << k, (uint32_t)MFM_CHUNK));
        if (mf_occ > 0) q_bytes = std::max(q_bytes, (size_t)(((158 * 1024 / mf_occ) - 1024) & ~255));
        if (q_bytes > 64 * 1024) ZK_TRY(c->allow_big_lds((const void*)multifold_mfma_kernel<4, 4>, 158 * 1024));   // (the kernel has ~0.6 KiB of static LDS on top)
        hipLaunchKernelGGL((multifold_mfma_kernel<4, 4>), dim3(tiles / 4), dim3(256), q_bytes, stream, cur, m, k, d_w, dst, pdst, rot);
    } else if (m >= 8192) {   // streaming shape: 64 outputs per workgroup, its waves split the terms
        out_per_wg = 64;
        // >= 64 terms per lane: every lane pays one 9-word reduction (~a product), which at 16 terms per lane made the
        // 6-variable fold of the overlapped plan 12 % slower than the 8-variable one
        const uint32_t waves = k >= 8 ? 4 : k == 7 ? 2 : 1;
        ProfScope ps(c, "multifold", 32.0 * (double)cn + 32.0 * (double)m, stream);
        hipLaunchKernelGGL((multifold_kernel<64, 4, true>), dim3((unsigned)(m / 64)), dim3(64 * waves), ((size_t)32 << k) + 32 * (size_t)(waves - 1) * 64,
                           stream, cur, m, k, d_w, dst, pdst);
    } else {                  // few outputs left: 16 per workgroup, up to 64 lanes share one output
        out_per_wg = 16;
        uint32_t waves = 16;
        while (waves * 4 > (1u << k)) waves >>= 1;   // at least one term per lane group (k >= 3 here)
        ProfScope ps(c, "multifold_small", 32.0 * (double)cn + 32.0 * (double)m, stream);
        hipLaunchKernelGGL(multifold_kernel<16>, dim3((unsigned)(m / 16)), dim3(64 * waves), ((size_t)32 << k) + 32 * (size_t)(waves - 1) * 16, stream, cur, m, k, d_w, dst, pdst);
    }
    ZK_HIP(c, hipGetLastError());
    if (n_parts) *n_parts = (uint32_t)(m / out_per_wg);
    return ZKHIP_OK;
}
// k-variable fold of a small table spread over the chip: partial tables P[y][m], y < *n_slices (blockfold_kernel)
struct BlockfoldShape { uint32_t log_ow, per, ny; };
// OW outputs x (1024 / OW) term slices per workgroup; at least one term per slice; ny partial tables come out
static inline bool blockfold_shape(uint32_t m, uint32_t k, BlockfoldShape* sh) {
    const uint32_t log_m = log2_exact(m);
    sh->log_ow = std::max<uint32_t>(std::min<uint32_t>(log_m, 5), k < 10 ? 10 - k : 0);
    if (sh->log_ow > log_m) return false;
    const uint32_t sl_cnt = (uint32_t)BF_BLOCK >> sh->log_ow;
    const uint32_t terms = 1u << k;
    sh->per = std::min<uint32_t>(4, terms / sl_cnt);
    sh->ny = terms / (sh->per * sl_cnt);
    return true;
}
static int launch_blockfold(zkhip_ctx* c, hipStream_t stream, const uint64_t* in, uint32_t m, uint32_t k, const uint64_t* d_w,
                            uint64_t* d_partial, uint32_t* n_slices) {
    BlockfoldShape sh;
    if (!blockfold_shape(m, k, &sh)) return ZKHIP_ERR_SHAPE;
    const uint32_t log_ow = sh.log_ow, per = sh.per, ny = sh.ny, terms = 1u << k;
    ProfScope ps(c, "blockfold", 32.0 * (double)m * terms, stream);
    hipLaunchKernelGGL(blockfold_kernel, dim3(m >> log_ow, ny), dim3(BF_BLOCK), 0, stream, in, m, log_ow, per, d_w, d_partial);
    ZK_HIP(c, hipGetLastError());
    *n_slices = ny;
    return ZKHIP_OK;
}
// fine block sums: d_fine[2^lb] = sums of the 2^lb equal consecutive blocks of the table (each >= FINE_CHUNK entries);
// d_chunk (n / FINE_CHUNK entries of scratch) is only touched when a block is longer than FINE_CHUNK
static int launch_fine_sums(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint32_t lb, uint64_t* d_fine, uint64_t* d_chunk,
                            hipStream_t stream = nullptr) {
    if (!stream) stream = c->stream;
    const size_t blk = n >> lb, n_chunks = n / FINE_CHUNK;
    uint64_t* first = blk == (size_t)FINE_CHUNK ? d_fine : d_chunk;
    {
        // (measured and dropped: waves that stay and loop over run pairs with the next pair's loads in flight -- 100-115 us
        // against 94-96 us for this one-shot form at 2^24; the loads alone take 81 us, tools/ubench_rows.hip "pieces")
        ProfScope ps(c, "fine_sums", 32.0 * (double)n, stream);
        // TWO workgroups (8 waves) per CU, through an LDS request the kernel never touches: with all the waves its registers allow
        // (6 per SIMD) the pass keeps ~48 MiB of loads in flight -- no faster (93 us against 90-91 us with two workgroups per CU, one:
        // 123 us), and every other kernel's loads queue behind them: the first rounds and small folds of the proofs in flight beside
        // it took 50-150 us instead of 10-40, 0.212 against 0.201 ms per proof (profiles/r06/NOTES.md section 7).
        // ZKHIP_FINE_LDS=<bytes> overrides (diagnostics; 0 = no cap).
        static const long fine_lds = [] { const char* e = getenv("ZKHIP_FINE_LDS"); return e ? atol(e) : 79872L; }();
        const size_t lds = (size_t)std::min<long>(std::max<long>(fine_lds, 0), 158 * 1024);
        if (lds > 64 * 1024) ZK_TRY(c->allow_big_lds((const void*)fine_sums_kernel, 158 * 1024));
        hipLaunchKernelGGL(fine_sums_kernel, dim3((unsigned)((n_chunks + 7) / 8)), dim3(MLE_BLOCK), lds, stream, d_evals, n_chunks, first);
    }
    if (first != d_fine)
        hipLaunchKernelGGL(group_sums_wg_kernel, dim3(1u << lb), dim3(MLE_BLOCK), 0, stream, first, (uint32_t)(blk / FINE_CHUNK), d_fine, (uint64_t*)nullptr);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

// (2^log_blocks block sums, then the total) of a table, on the device
static int block_sums_impl(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint32_t log_blocks, uint64_t* d_out, uint64_t* h_total, bool want_total);
extern "C" int zkhip_mle_block_sums(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint32_t log_blocks,
                                    uint64_t* d_out, uint64_t* h_total) {
    return block_sums_impl(c, d_evals, n, log_blocks, d_out, h_total, true);
}
// The same without the total where it would cost a launch of its own (the fine granularity of the overlapped plan): a prover that
// is going to absorb the TRUE sum gets it from its own sum tree (zkhip_sumcheck_prove with both claimed-sum arguments NULL), and
// zkhip_mle_block_sums_total delivers it to a caller who wants to look at it first.
extern "C" int zkhip_mle_block_sums_deferred(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint32_t log_blocks, uint64_t* d_out) {
    return block_sums_impl(c, d_evals, n, log_blocks, d_out, nullptr, false);
}
extern "C" int zkhip_mle_block_sums_total(zkhip_ctx* c, uint64_t* d_block_sums, uint32_t log_blocks, uint64_t* h_total) {
    if (!c || !d_block_sums || !h_total || log_blocks > 16) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    // two levels: 2^(lb/2) workgroups sum runs of the block sums into the context's scratch, one workgroup adds those
    const uint32_t hi = log_blocks / 2, lo = log_blocks - hi;
    uint64_t* tmp = c->small_u64(ZK_SMALL_PARTIALS);
    hipLaunchKernelGGL(group_sums_wg_kernel, dim3(1u << hi), dim3(MLE_BLOCK), 0, c->stream, d_block_sums, 1u << lo, tmp, (uint64_t*)nullptr);
    hipLaunchKernelGGL(group_sums_wg_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, tmp, 1u << hi, d_block_sums + 4 * ((size_t)1 << log_blocks), (uint64_t*)nullptr);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_block_sums + 4 * ((size_t)1 << log_blocks), 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_total, c->pinned_u64(ZK_PIN_RES), 32);
    return ZKHIP_OK;
}
static int block_sums_impl(zkhip_ctx* c, const uint64_t* d_evals, size_t n, uint32_t log_blocks, uint64_t* d_out, uint64_t* h_total, bool want_total) {
    if (!c || !d_evals || !d_out) return ZKHIP_ERR_ARG;
    if (!is_pow2(n) || log_blocks > 16 || ((size_t)1 << log_blocks) > n) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    const size_t m = n >> log_blocks;
    if (log_blocks > (uint32_t)MF_CAP_LOGK) {      // fine sums (the overlapped prover's granularity)
        if (m < (size_t)FINE_CHUNK) return ZKHIP_ERR_SHAPE;
        uint64_t* d_chunk = nullptr;
        if (m > (size_t)FINE_CHUNK) { ZK_TRY(c->reserve_ws((n / FINE_CHUNK) * 32)); d_chunk = (uint64_t*)c->d_ws; }
        ZK_TRY(launch_fine_sums(c, d_evals, n, log_blocks, d_out, d_chunk));
        // With the total: coarse sums at the granularity the prover's first rounds want (kept for it in the context's ring), then the
        // total from those.  Without (the deferred form): nothing more here -- the prover derives the coarse sums itself, on the
        // stream of the proof (a high-priority one when the proof is in flight), instead of queueing a tiny kernel on the caller's
        // stream behind whatever streaming pass occupies the chip.
        // (Round 3 tried to have the sums pass deliver the coarse sums too and dropped both forms.  Per-workgroup sums (n / 2048 values)
        // added up by the first serial kernel's prologue: the 8 us kernel and its 6 us launch gap went, the prologue grew by 9 us and the
        // step stayed at 0.334 ms.  The workgroup that finishes LAST under a coarse block adds them up (a device-scope release + counter
        // per workgroup): that release writes the L2 back once per workgroup, 8192 times -- 0.75 ms per step instead of 0.33.)
        if (want_total) {
            const uint32_t k1 = overlapped_plan(n) && log_blocks == log2_exact(n) - 8 ? log_blocks - overlapped_k2(n) : 8;
            int cs = 0;
            ZK_TRY(c->next_coarse(&cs));
            uint64_t* coarse_canon = (uint64_t*)c->d_coarse[cs], *coarse_mont = coarse_canon + 4 * 1024;
            {
                ProfScope ps(c, "coarse_sums", 0.0);
                hipLaunchKernelGGL(group_sums_wg_kernel, dim3(1u << k1), dim3(MLE_BLOCK), 0, c->stream, d_out, 1u << (log_blocks - k1), coarse_mont, coarse_canon);
            }
            {
                ProfScope ps(c, "total_sum", 0.0);
                hipLaunchKernelGGL(group_sums_wg_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, coarse_mont, 1u << k1, d_out + 4 * ((size_t)1 << log_blocks), (uint64_t*)nullptr);
            }
            c->coarse_of[cs] = d_out; c->coarse_n[cs] = n; c->coarse_k1[cs] = k1;
        }
    } else {
        const uint32_t chunk = (uint32_t)std::min<size_t>(m, 4096);
        const size_t n_chunks = n / chunk;
        uint64_t* d_partials;
        if (n_chunks <= 8 * (size_t)ZK_MAX_PARTIALS) d_partials = c->small_u64(ZK_SMALL_PARTIALS);
        else { ZK_TRY(c->reserve_ws(n_chunks * 32)); d_partials = (uint64_t*)c->d_ws; }
        if (chunk >= (uint32_t)MLE_BLOCK) {
            ProfScope ps(c, "chunk_sums", 32.0 * (double)n);
            hipLaunchKernelGGL(chunk_sums_kernel, dim3((unsigned)n_chunks), dim3(MLE_BLOCK), 0, c->stream, d_evals, chunk, d_partials);
            hipLaunchKernelGGL(group_sums_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_partials, (uint32_t)(m / chunk),
                               1u << log_blocks, d_out);
        } else {   // tiny table: every entry is its own partial
            hipLaunchKernelGGL(group_sums_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_evals, (uint32_t)m, 1u << log_blocks, d_out);
        }
    }
    ZK_HIP(c, hipGetLastError());
    if (h_total) {
        ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_out + 4 * ((size_t)1 << log_blocks), 32, hipMemcpyDeviceToHost, c->stream));
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        std::memcpy(h_total, c->pinned_u64(ZK_PIN_RES), 32);
    }
    return ZKHIP_OK;
}

// The prover in two halves: sumcheck_enqueue launches every kernel and the copy of the proof into pinned slot `slot`,
// sumcheck_collect waits for that copy (polling its event) and hands the proof out.  A caller with several tables to prove
// begins the next proof before it collects the previous one (zkhip_sumcheck_prove_begin / _end): the kernels of successive
// proofs run in stream order, what disappears is the idle time between them (host wake-up, return, next call's first launch).
// `lane` < 0: on the caller's stream with the context's buffers (the synchronous call); otherwise on the private streams and buffers of
// c->lanes[lane], behind everything the caller's stream holds at this moment
static int sumcheck_enqueue(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_claimed_sum,
                            const uint64_t* d_claimed_sum, const uint64_t* d_block_sums, uint32_t log_blocks, int slot, int lane, int in_flight = 1) {
    const uint32_t n_vars = log2_exact(n);
    const bool overlap = overlapped_plan(n);
    // workspace: stage tables (n/4 + n/16 + ...; overlapped: n / 2^k1 <= n/8), partial sums, fold weights, and for the
    // overlapped plan the fine sums (n/256 <= 2^16) and two sets of partial tables (<= 2 x 1024 and 32 x 256)
    const size_t tab_entries = overlap ? n / 8 + 64 : n / 4 + n / 16 + 64;
    const size_t tabA_entries = overlap ? n / 8 + 32 : n / 4 + 32;
    const size_t part_entries = std::max<size_t>(n / 4096, n / 256) + 1024;   // chunk sums of stage 0, per-workgroup sums of stage outputs
    const size_t w_entries = (size_t)1 << TREE_MAX_LOG;
    const size_t fine_entries = overlap ? 65536 + 2 * 1024 + 32 * 256 : 0;
    const size_t ws_need = (tab_entries + 2 * part_entries + 2 * w_entries + fine_entries) * 32;
    hipStream_t S, F = nullptr;                   // serial / fold streams of this proof
    hipEvent_t fork_ev = nullptr, serial_ev = nullptr;
    uint64_t *ws, *small;
    if (lane < 0) {
        ZK_TRY(c->reserve_ws(ws_need));
        ws = (uint64_t*)c->d_ws; small = (uint64_t*)c->d_small; S = c->stream;
        if (overlap) { ZK_TRY(c->ensure_fold_stream()); F = c->fold_stream; fork_ev = c->fork_ev; serial_ev = c->serial_ev; }
    } else {
        ZK_TRY(c->ensure_lane(lane, ws_need));
        zkhip_ctx::ProofLane& L = c->lanes[lane];
        ws = (uint64_t*)L.ws; small = (uint64_t*)L.small; S = L.serial; F = L.fold; fork_ev = L.fork_ev; serial_ev = L.serial_ev;
        ZK_HIP(c, hipEventRecord(L.begin_ev, c->stream));          // the table, its block sums and the claimed sum are ready behind this
        ZK_HIP(c, hipStreamWaitEvent(S, L.begin_ev, 0));
    }
    uint64_t* tabA = ws;
    uint64_t* tabB = tabA + 4 * tabA_entries;
    uint64_t* partA = ws + 4 * tab_entries;
    uint64_t* partB = partA + 4 * part_entries;
    uint64_t* d_w = partB + 4 * part_entries;
    uint64_t* d_w2 = d_w + 4 * w_entries;
    uint64_t* d_fine = d_w2 + 4 * w_entries;           // overlapped plan only from here on
    uint64_t* d_p1 = d_fine + 4 * 65536;
    uint64_t* d_p2 = d_p1 + 4 * 2 * 1024;
    SumcheckDev* st = (SumcheckDev*)(small + ZK_SMALL_STATE);
    uint64_t* d_rp = small + ZK_SMALL_ROUNDPOLYS;
    uint64_t* d_ch = small + ZK_SMALL_CHALLENGES;
    uint64_t* d_fin = small + ZK_SMALL_RES;
    // the proof leaves through the LAST serial kernel: it writes [sum .. round polynomials] into the pinned slot itself (SmallArgs::host_delta)
    ZK_TRY(c->ensure_proof_slot(slot));
    const long long host_delta = ((long long)(intptr_t)c->proof_pin[slot] - (long long)(intptr_t)(small + ZK_SMALL_STATE)) / 8;

    FrArg claimed = {};
    uint32_t first = 1;
    if (h_claimed_sum) { std::memcpy(claimed.v, h_claimed_sum, 32); first = 2; }   // prove(&self) absorbs self.sum (sumcheck.rs:33-35)
    else if (d_claimed_sum) first = 3;

    const uint64_t* cur = d_evals;
    size_t cn = n;
    uint32_t round = 0, stage = 0;
    const uint64_t* parts = nullptr;   // partial sums of `cur`, `group` consecutive ones per block of this stage
    uint32_t n_parts = 0;
    bool done = false;
    hipStream_t tail_stream = S;              // where the last kernels of the proof run (the overlapped plan ends on its fold stream)
    if (overlap) {
        const uint32_t g = n_vars - 8, k2 = overlapped_k2(n), k1 = g - k2;
        const uint64_t* fine = d_block_sums;
        if (!fine || log_blocks != g) {             // poly_sum() was not called (or with another granularity)
            // three or more proofs in flight: this streaming pass belongs onto the caller's stream like every other one (below); the
            // serial stream follows it
            const bool on_callers = lane >= 0 && in_flight >= 3;
            ZK_TRY(launch_fine_sums(c, d_evals, n, g, d_fine, nullptr, on_callers ? c->stream : S));
            if (on_callers) {
                ZK_HIP(c, hipEventRecord(c->lanes[lane].begin_ev, c->stream));
                ZK_HIP(c, hipStreamWaitEvent(S, c->lanes[lane].begin_ev, 0));
            }
            fine = d_fine;
        }
        // the coarse sums poly_sum() left for this table (the newest ring entry that names its fine sums), or our own
        int cs = -1;
        for (int q = 1; q <= zkhip_ctx::COARSE_RING && cs < 0; ++q) {
            const int e = (c->coarse_next - q + 2 * zkhip_ctx::COARSE_RING) % zkhip_ctx::COARSE_RING;
            if (fine == d_block_sums && c->coarse_of[e] == fine && c->coarse_n[e] == n && c->coarse_k1[e] == k1) cs = e;
        }
        if (cs < 0) {
            ZK_TRY(c->next_coarse(&cs));
            ProfScope ps(c, "coarse_sums", 0.0, S);
            hipLaunchKernelGGL(group_sums_wg_kernel, dim3(1u << k1), dim3(MLE_BLOCK), 0, S, fine, 1u << k2, (uint64_t*)nullptr, (uint64_t*)c->d_coarse[cs]);
        }
        const uint64_t* coarse = (const uint64_t*)c->d_coarse[cs];
        c->coarse_of[cs] = nullptr;                 // the entry belongs to this proof from here on:
        c->coarse_owner[cs] = slot + 1;             // the ring skips it until the proof has been collected (sumcheck_collect / prove_end)
        SmallArgs a = {};
        a.src = coarse; a.group = 0; a.canon = 1; a.log_n = k1; a.n_rounds = k1; a.round0 = 0; a.first = first; a.claimed = claimed;
        a.d_claimed = d_claimed_sum; a.weights_out = d_w; a.final_out = nullptr;
        ZK_TRY(launch_small(c, a, st, d_rp, d_ch, 0, S));
        first = 0;
        // fork: the big fold on the fold stream next to rounds k1+1 .. g.  The serial kernel of those rounds asks for
        // (nearly) a whole CU's LDS, so no fold workgroup shares its CU (beside 8+ fold waves per SIMD the transcript
        // wave ran at half speed)
        // The fork comes BEHIND the small fold: the serial kernel (next in this queue) and the big fold (another queue, behind
        // an event) then become ready together and the serial kernel's single workgroup is placed first.  Forked before
        // the small fold, the big fold filled every CU first and the serial kernel waited for one to drain (~60 us).
        uint32_t ny = 0;
        ZK_TRY(launch_blockfold(c, S, fine, 1u << k2, k1, d_w, d_p1, &ny));
        ZK_HIP(c, hipEventRecord(fork_ev, S));
        SmallArgs b = {};
        b.src = d_p1; b.group = ny; b.stride = 1u << k2; b.canon = 1; b.log_n = k2; b.n_rounds = k2; b.round0 = k1; b.first = 0;
        b.weights_out = d_w2; b.final_out = nullptr;
        if (lane >= 0 && in_flight < 3) { ZK_TRY(c->ensure_lane_fold(lane)); F = c->lanes[lane].fold; }
        if (lane >= 0 && in_flight >= 3) {
            // Three or more proofs in flight: the streaming passes of ALL of them on ONE stream, the caller's -- where poly_sum() puts the sums
            // passes already.  The big fold and what follows it are enqueued LATER (zkhip_ctx::deferred), behind the sums passes of the next
            // one to three tables, so that the caller's stream never stands waiting for this proof's first rounds; the serial kernel of
            // rounds k1+1 .. g goes out now, the last stage follows the fold on the serial stream.  Measured at 2^24 (profiles/r06/NOTES.md
            // section 7): 0.207 ms per proof with eight in flight against 0.233 with a fold stream per lane.
            ZK_TRY(launch_small(c, b, st, d_rp, d_ch, 156 * 1024, S));
            hipEvent_t fold_ev = serial_ev;
            c->deferred_rc[slot] = ZKHIP_OK;
            c->deferred.emplace_back(slot, [=]() -> int {
                uint32_t np2 = 0, ny2 = 0;
                ZK_HIP(c, hipStreamWaitEvent(c->stream, fork_ev, 0));
                ZK_TRY(launch_multifold(c, c->stream, d_evals, n, k1, d_w, tabA, partA, &np2));
                ZK_HIP(c, hipEventRecord(fold_ev, c->stream));
                ZK_HIP(c, hipStreamWaitEvent(S, fold_ev, 0));
                ZK_TRY(launch_blockfold(c, S, tabA, 256, k2, d_w2, d_p2, &ny2));
                SmallArgs t = {};
                t.src = d_p2; t.group = ny2; t.stride = 256; t.canon = 1; t.log_n = 8; t.n_rounds = 8; t.round0 = g; t.first = 0;
                t.weights_out = nullptr; t.final_out = d_fin; t.host_delta = host_delta;
                ZK_TRY(launch_small(c, t, st, d_rp, d_ch, 0, S));
                ZK_HIP(c, hipEventRecord(c->proof_ev[slot], S));
                return ZKHIP_OK;
            });
            // how many second halves stay back: enough sums passes in front of a fold that its proof's first rounds are over when its turn
            // comes (1-3 tables), few enough that the host, which stops at its depth, still has passes queued (depth 3 / 4: 1, 5: 2, 6+: 3)
            // (a second half that cannot be enqueued is its OWN proof's failure -- recorded in deferred_rc, reported by that proof's prove_end --
            // not this proof's: this one's second half is in the queue now and the ticket must reach the caller)
            (void)c->flush_deferred((size_t)std::min(3, std::max(1, in_flight - 3)));
            return ZKHIP_OK;
        }
        ZK_HIP(c, hipStreamWaitEvent(F, fork_ev, 0));
        ZK_TRY(launch_multifold(c, F, d_evals, n, k1, d_w, tabA, partA, &n_parts));
        ZK_TRY(launch_small(c, b, st, d_rp, d_ch, 156 * 1024, S));
        // join ON THE FOLD STREAM: the serial kernel ends well before the big fold, so its event is long set when the fold
        // ends and the last stage follows the fold in stream order (joining on the caller's stream left the chip idle for the
        // ~13 us a cross-stream dependency takes to resolve); the proof is copied from there too
        ZK_HIP(c, hipEventRecord(serial_ev, S));
        ZK_HIP(c, hipStreamWaitEvent(F, serial_ev, 0));
        tail_stream = F;
        round = g;
        ZK_TRY(launch_blockfold(c, tail_stream, tabA, 256, k2, d_w2, d_p2, &ny));       // 2^(8 + k2) entries -> 2^8, the last 8 rounds
        SmallArgs t = {};
        t.src = d_p2; t.group = ny; t.stride = 256; t.canon = 1; t.log_n = 8; t.n_rounds = 8; t.round0 = round; t.first = 0;
        t.weights_out = nullptr; t.final_out = d_fin; t.host_delta = host_delta;
        ZK_TRY(launch_small(c, t, st, d_rp, d_ch, 0, tail_stream));
        done = true;
    }
    while (!done && stage_k(cn) != 0) {
        const uint32_t k = stage_k(cn);
        const size_t m = cn >> k;
        SmallArgs a = {};
        if (stage == 0 && d_block_sums && log_blocks >= k && log_blocks <= (uint32_t)MF_CAP_LOGK) {   // poly_sum() already streamed the table once
            a.src = d_block_sums;
            a.group = 1u << (log_blocks - k);
        } else if (stage == 0) {
            const uint32_t chunk = (uint32_t)std::min<size_t>(m, 4096);   // m >= 256 here
            ProfScope ps(c, "chunk_sums", 32.0 * (double)cn, S);
            hipLaunchKernelGGL(chunk_sums_kernel, dim3((unsigned)(cn / chunk)), dim3(MLE_BLOCK), 0, S, cur, chunk, partA);
            a.src = partA;
            a.group = (uint32_t)(m / chunk);
        } else {
            a.src = parts;
            a.group = (uint32_t)(n_parts >> k);
        }
        a.log_n = k; a.n_rounds = k; a.round0 = round; a.first = first; a.claimed = claimed; a.d_claimed = d_claimed_sum;
        a.weights_out = d_w; a.final_out = nullptr;
        ZK_TRY(launch_small(c, a, st, d_rp, d_ch, 0, S));
        first = 0;
        uint64_t* dst = (stage & 1) ? tabB : tabA;
        uint64_t* pdst = (stage & 1) ? partB : partA;
        ZK_TRY(launch_multifold(c, S, cur, cn, k, d_w, dst, pdst, &n_parts));
        parts = pdst;
        cur = dst;
        cn = m;
        round += k;
        ++stage;
    }
    if (!done) {
        SmallArgs a = {};
        a.src = cur; a.group = 0; a.log_n = log2_exact(cn); a.n_rounds = a.log_n; a.round0 = round; a.first = first;
        a.claimed = claimed; a.d_claimed = d_claimed_sum; a.weights_out = nullptr; a.final_out = d_fin; a.host_delta = host_delta;
        ZK_TRY(launch_small(c, a, st, d_rp, d_ch, 0, S));
    }
    // results -> host: written by the last kernel into the pinned slot (no copy launch); the collector polls the event behind it
    // the event the collector polls; the caller's stream stays ordered behind the proof (the next call reuses the scratch)
    ZK_HIP(c, hipEventRecord(c->proof_ev[slot], tail_stream));
    // (a proof on a lane of its own is joined to the caller's stream when it is collected, so that the caller's next poly_sum() does not wait for it)
    if (lane < 0 && tail_stream != c->stream) ZK_HIP(c, hipStreamWaitEvent(c->stream, c->proof_ev[slot], 0));
    return ZKHIP_OK;
}
static int sumcheck_collect(zkhip_ctx* c, int slot, uint32_t n_vars, uint64_t* h_sum, uint64_t* h_round_polys, uint64_t* h_challenges) {
    const int wrc = c->wait_event(c->proof_ev[slot]);
    c->release_coarse(slot);                    // the proof's kernels are done with their coarse sums (or the device is lost)
    ZK_TRY(wrc);
    const uint64_t* pin = (const uint64_t*)c->proof_pin[slot];
    const uint64_t* span = c->small_u64(ZK_SMALL_STATE);
    const SumcheckDev* st = (const SumcheckDev*)c->small_u64(ZK_SMALL_STATE);
    std::memcpy(h_sum, pin + ((const uint64_t*)st->sum - span), 32);
    std::memcpy(h_round_polys, pin + (c->small_u64(ZK_SMALL_ROUNDPOLYS) - span), 64 * (size_t)n_vars);
    std::memcpy(h_challenges, pin + (c->small_u64(ZK_SMALL_CHALLENGES) - span), 32 * (size_t)n_vars);
    return ZKHIP_OK;
}
static int sumcheck_check_args(zkhip_ctx* c, const uint64_t* d_evals, size_t n) {
    if (!c || !d_evals) return ZKHIP_ERR_ARG;
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;   // Multilinear::new evaluation_form.rs:16-20
    if (log2_exact(n) > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    return ZKHIP_OK;
}

extern "C" int zkhip_sumcheck_prove(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_claimed_sum,
                                    const uint64_t* d_claimed_sum, const uint64_t* d_block_sums, uint32_t log_blocks,
                                    uint64_t* h_sum,
                                    uint64_t* h_round_polys, uint64_t* h_challenges) {
    ZK_TRY(sumcheck_check_args(c, d_evals, n));
    if (!h_sum) return ZKHIP_ERR_ARG;
    const uint32_t n_vars = log2_exact(n);
    if (n_vars && (!h_round_polys || !h_challenges)) return ZKHIP_ERR_ARG;
    for (int k = 0; k < zkhip_ctx::PROOF_SLOTS; ++k) if (c->proof_pending[k]) return ZKHIP_ERR_BUSY;     // proofs in flight own the result slots
    ZK_TRY(c->activate());
    if (n == 1) {   // no rounds: nothing is proven; report the sum the transcript would have absorbed
        if (h_claimed_sum) { std::memcpy(h_sum, h_claimed_sum, 32); return ZKHIP_OK; }
        ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_claimed_sum ? d_claimed_sum : d_evals, 32, hipMemcpyDeviceToHost, c->stream));
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        std::memcpy(h_sum, c->pinned_u64(ZK_PIN_RES), 32);
        return ZKHIP_OK;
    }
    ZK_TRY(sumcheck_enqueue(c, d_evals, n, h_claimed_sum, d_claimed_sum, d_block_sums, log_blocks, 0, -1));
    return sumcheck_collect(c, 0, n_vars, h_sum, h_round_polys, h_challenges);
}
// Sumcheck::prove in flight: begin enqueues the whole proof and returns a ticket, end waits for it and delivers the outputs of
// zkhip_sumcheck_prove.  Up to eight proofs of tables with >= 2 entries may be in flight, each on streams and buffers of its own
// (zkhip_ctx::ProofLane): the streaming passes of one run while the transcript rounds of the others hash.
extern "C" int zkhip_sumcheck_prove_begin(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_claimed_sum,
                                          const uint64_t* d_claimed_sum, const uint64_t* d_block_sums, uint32_t log_blocks,
                                          uint32_t* ticket) {
    ZK_TRY(sumcheck_check_args(c, d_evals, n));
    if (!ticket || n < 2) return ZKHIP_ERR_ARG;
    int slot = -1;
    for (int k = 0; k < zkhip_ctx::PROOF_SLOTS; ++k) if (!c->proof_pending[k]) { slot = k; break; }
    if (slot < 0) return ZKHIP_ERR_BUSY;
    ZK_TRY(c->activate());
    int in_flight = 1;
    for (int k = 0; k < zkhip_ctx::PROOF_SLOTS; ++k) in_flight += c->proof_pending[k] != 0;
    ZK_TRY(sumcheck_enqueue(c, d_evals, n, h_claimed_sum, d_claimed_sum, d_block_sums, log_blocks, slot, slot, in_flight));
    c->proof_pending[slot] = log2_exact(n);
    *ticket = (uint32_t)slot;
    return ZKHIP_OK;
}
extern "C" int zkhip_sumcheck_prove_end(zkhip_ctx* c, uint32_t ticket, uint64_t* h_sum, uint64_t* h_round_polys, uint64_t* h_challenges) {
    if (!c || ticket >= (uint32_t)zkhip_ctx::PROOF_SLOTS || !c->proof_pending[ticket]) return ZKHIP_ERR_ARG;
    const uint32_t n_vars = c->proof_pending[ticket];
    int rc = c->activate();
    if (rc == ZKHIP_OK) {
        (void)c->flush_deferred(0, (int)ticket);          // the proof's second half, if it is still held back (and those of the older proofs in front of it)
        rc = c->deferred_rc[ticket];
        c->deferred_rc[ticket] = ZKHIP_OK;
        if (rc != ZKHIP_OK) {                   // the proof never got its last stage: drain what did get out, there is nothing to collect
            (void)hipStreamSynchronize(c->lanes[ticket].serial);
            (void)hipStreamSynchronize(c->stream);
            c->release_coarse((int)ticket);
        }
    }
    if (rc == ZKHIP_OK) {
        if (h_sum && h_round_polys && h_challenges) rc = sumcheck_collect(c, (int)ticket, n_vars, h_sum, h_round_polys, h_challenges);
        else { rc = c->wait_event(c->proof_ev[ticket]); c->release_coarse((int)ticket); }   // abandoned: just wait it out
        if (hipStreamWaitEvent(c->stream, c->proof_ev[ticket], 0) != hipSuccess && rc == ZKHIP_OK) rc = ZKHIP_ERR_HIP;   // the caller's stream is ordered behind the proof again
    }
    c->proof_pending[ticket] = 0;
    return rc;
}

// ---------------------------------------------------------------------------------------
// split-phase prover for a table sharded over several GPUs
// ---------------------------------------------------------------------------------------
struct zkhip_sc_state {
    zkhip_ctx* c;
    const uint64_t* cur;     // current local table
    size_t cn;               // its length
    uint64_t *A, *B;         // ping-pong buffers (n/2, n/4 entries)
    uint64_t* small;         // [SumcheckDev 64 u64][round polys 8*R][challenges 4*R][partials 8*MLE_MAX_GRID][tail 4*TAIL_N]
    uint32_t round, np;
    bool partials_valid;
    bool owns_tables;        // false: A/B live in the context workspace (the common, single-state case)
    bool uses_cache;         // small + stage buffers borrowed from the context
    // stage form
    uint64_t* stage_buf;     // [weights 4*512][partials X 4*P][partials Y 4*P][block sums 4*513]
    size_t stage_parts_cap;
    uint32_t stage_k_cur, stage_world, stage_idx, n_parts;
    const uint64_t* parts;
    // overlapped stage (zkhip_sc_overlap_*): scratch inside B -- fine sums, fold weights of both halves, partial tables, local table
    uint32_t ov_k1 = 0, ov_k2 = 0, ov_ny1 = 0, ov_phase = 0;
    uint64_t* ov_fine() { return B; }
    uint64_t* ov_w1() { return B + 4 * (size_t)65536; }
    uint64_t* ov_w2() { return ov_w1() + 4 * ((size_t)1 << TREE_MAX_LOG); }
    uint64_t* ov_p2() { return ov_w2() + 4 * ((size_t)1 << TREE_MAX_LOG); }      // <= 32 x 256
    uint64_t* ov_loc() { return ov_p2() + 4 * (size_t)32 * 256; }                // 256
    uint64_t* sw() { return stage_buf; }
    uint64_t* spx() { return stage_buf + 4 * ((size_t)1 << MF_CAP_LOGK); }
    uint64_t* spy() { return spx() + 4 * stage_parts_cap; }
    uint64_t* sbs() { return spy() + 4 * stage_parts_cap; }
    SumcheckDev* dev() { return (SumcheckDev*)small; }
    uint64_t* rp() { return small + 64; }
    uint64_t* ch() { return rp() + 8 * ZK_MAX_ROUNDS; }
    uint64_t* partials() { return ch() + 4 * ZK_MAX_ROUNDS; }
    uint64_t* fin() { return partials() + 8 * (size_t)MLE_MAX_GRID; }
};

extern "C" int zkhip_sc_begin(zkhip_ctx* c, const uint64_t* d_local, size_t n_local, zkhip_sc_state** out) {
    if (!c || !d_local || !out) return ZKHIP_ERR_ARG;
    if (!is_pow2(n_local)) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    zkhip_sc_state* st = new zkhip_sc_state();
    st->c = c; st->cur = d_local; st->cn = n_local; st->round = 0; st->np = 0; st->partials_valid = false;
    st->A = st->B = st->small = nullptr;
    const size_t small_u64 = 64 + 12 * (size_t)ZK_MAX_ROUNDS + 8 * (size_t)MLE_MAX_GRID + 16;
    st->stage_parts_cap = std::max<size_t>(n_local / 256, 64) + 64;
    st->uses_cache = false;
    if (!c->sc_lent && c->sc_small && c->sc_stage_cap >= st->stage_parts_cap) {
        st->small = (uint64_t*)c->sc_small;
        st->stage_buf = (uint64_t*)c->sc_stage;
        st->stage_parts_cap = c->sc_stage_cap;
        st->uses_cache = true;
        c->sc_lent = true;
    } else {
        if (hipMalloc(&st->small, small_u64 * 8) != hipSuccess) { delete st; return ZKHIP_ERR_NOMEM; }
        st->stage_buf = nullptr;
        if (hipMalloc(&st->stage_buf, (((size_t)1 << MF_CAP_LOGK) + 2 * st->stage_parts_cap + ((size_t)1 << MF_CAP_LOGK) + 1 + 8) * 32) != hipSuccess) {
            hipFree(st->small); delete st; return ZKHIP_ERR_NOMEM;
        }
    }
    st->owns_tables = false;
    if (n_local >= 2) {
        const size_t bytes = (n_local / 2 + n_local / 4 + 4) * 32;
        if (!c->ws_lent) {               // steady state: no allocation per prove
            int rc = c->reserve_ws(bytes);
            if (rc != ZKHIP_OK) { if (st->uses_cache) c->sc_lent = false; else { hipFree(st->small); hipFree(st->stage_buf); } delete st; return rc; }
            st->A = (uint64_t*)c->d_ws;
            c->ws_lent = true;
        } else {                         // several states alive at once (tests drive shards in lockstep)
            if (hipMalloc(&st->A, bytes) != hipSuccess) { if (st->uses_cache) c->sc_lent = false; else { hipFree(st->small); hipFree(st->stage_buf); } delete st; return ZKHIP_ERR_NOMEM; }
            st->owns_tables = true;
        }
    }
    st->B = st->A ? st->A + 4 * (n_local / 2) : nullptr;
    st->stage_k_cur = 0; st->stage_world = 1; st->stage_idx = 0; st->n_parts = 0; st->parts = nullptr;
    *out = st;
    return ZKHIP_OK;
}

// ---- stage form -------------------------------------------------------------------------------------
// The plans are PURE functions of (entries per shard, world): a rank that failed -- possibly before it had a state at all -- still
// walks the exchange schedule of the protocol with them (shard_protocol.hpp, "a failing rank must not hang its peers").
int zk_sc_plan_stage(size_t cn, uint32_t world, uint32_t* k_out) {
    if (!k_out || !is_pow2(world) || !is_pow2(cn)) return ZKHIP_ERR_ARG;
    uint32_t k;
    if (world == 1) {
        k = stage_k(cn);                        // the single-GPU plan
    } else {
        // Every stage costs an exchange: use as few as the kernels allow (<= MF_CAP_LOGK variables per stage, the
        // gathered tail takes zkhip_sc_tail_capacity() entries) and spread the variables evenly over them.
        const uint32_t lg = log2_exact(cn * world);
        const uint32_t tail_log = log2_exact((size_t)zkhip_sc_tail_capacity());
        if (lg <= tail_log) k = 0;
        else {
            const uint32_t need = lg - tail_log;
            const uint32_t stages = (need + MF_CAP_LOGK - 1) / MF_CAP_LOGK;
            k = std::max<uint32_t>((need + stages - 1) / stages, 3);   // a stage folds at least 3 variables (overshooting the tail size is fine)
        }
    }
    while (k && (cn >> k) < 16) --k;           // the local k-variable fold needs >= 16 outputs per workgroup
    if (k < 3) k = 0;                           // too little left: gather the tables and finish replicated
    *k_out = k;
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_stage_plan(zkhip_sc_state* st, uint32_t world, uint32_t* k_out) {
    if (!st || !k_out || !is_pow2(world)) return ZKHIP_ERR_ARG;
    uint32_t k = 0;
    ZK_TRY(zk_sc_plan_stage(st->cn, world, &k));
    st->stage_k_cur = k;
    st->stage_world = world;
    *k_out = k;
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_stage_block_sums(zkhip_sc_state* st, uint64_t* d_out) {
    if (!st || !d_out || st->stage_k_cur == 0) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    const uint32_t k = st->stage_k_cur;
    const size_t m = st->cn >> k;
    if (st->stage_idx == 0) {
        const uint32_t chunk = (uint32_t)std::min<size_t>(m, 4096);
        if (chunk >= (uint32_t)MLE_BLOCK) {
            ProfScope ps(c, "chunk_sums", 32.0 * (double)st->cn);
            hipLaunchKernelGGL(chunk_sums_kernel, dim3((unsigned)(st->cn / chunk)), dim3(MLE_BLOCK), 0, c->stream, st->cur, chunk, st->spx());
            hipLaunchKernelGGL(group_sums_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, st->spx(), (uint32_t)(m / chunk), 1u << k, st->sbs());
        } else {
            hipLaunchKernelGGL(group_sums_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, st->cur, (uint32_t)m, 1u << k, st->sbs());
        }
    } else {
        hipLaunchKernelGGL(group_sums_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, st->parts, st->n_parts >> k, 1u << k, st->sbs());
    }
    ZK_HIP(c, hipMemcpyAsync(d_out, st->sbs(), 32 * ((size_t)1 << k), hipMemcpyDeviceToDevice, c->stream));
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_stage_absorb(zkhip_sc_state* st, const uint64_t* d_gathered, uint32_t world, const uint64_t* h_claimed) {
    if (!st || !d_gathered || st->stage_k_cur == 0 || world != st->stage_world) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    const uint32_t k = st->stage_k_cur;
    SmallArgs a = {};
    a.src = d_gathered; a.group = world; a.stride = 1u << k;
    a.log_n = k; a.n_rounds = k; a.round0 = st->round;
    a.first = st->round == 0 ? 1u : 0u;
    if (a.first && h_claimed) { std::memcpy(a.claimed.v, h_claimed, 32); a.first = 2; }
    a.weights_out = st->sw(); a.final_out = nullptr;
    ZK_TRY(launch_small(c, a, st->dev(), st->rp(), st->ch()));
    st->round += k;
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_stage_fold(zkhip_sc_state* st) {
    if (!st || st->stage_k_cur == 0) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    const uint32_t k = st->stage_k_cur;
    const size_t m = st->cn >> k;
    uint64_t* dst = (st->cur == st->A) ? st->B : st->A;     // m <= n_local/8 fits either buffer
    uint64_t* pdst = (st->parts == st->spx() || st->stage_idx == 0) ? st->spy() : st->spx();
    uint32_t n_parts = 0;
    ZK_TRY(launch_multifold(c, c->stream, st->cur, st->cn, k, st->sw(), dst, pdst, &n_parts));
    st->parts = pdst;
    st->n_parts = n_parts;
    st->cur = dst;
    st->cn = m;
    st->partials_valid = false;
    st->stage_k_cur = 0;
    st->stage_idx++;
    return ZKHIP_OK;
}
// ---- overlapped stage: the single-GPU plan of 2^19..2^24-entry tables (sumcheck_enqueue) in exchange form --------------
// The local shard is summed once at FINE_CHUNK granularity.  Exchange 1 carries the 2^k1 coarse sums (k1 rounds on them,
// replicated); the fine sums folded by those k1 variables are the block sums of the next k2 rounds, exchange 2 carries them
// while the k1-variable fold of the shard runs on the context's fold stream; after the k2 rounds the fold's output is folded
// by the k2 variables into a 256-entry local table (gathered by the caller for the last rounds).  Three exchanges in all,
// as in the plain stage form, with the big fold hidden behind the second one and the serial kernel of the k2 rounds.
int zk_sc_plan_overlap(size_t cn, uint32_t world, uint32_t* k1, uint32_t* k2, uint32_t* mid_entries, uint32_t* ny_out) {
    if (!k1 || !k2 || !mid_entries || !is_pow2(world)) return ZKHIP_ERR_ARG;
    *k1 = *k2 = *mid_entries = 0;
    if (ny_out) *ny_out = 0;
    if (!overlapped_shard(cn)) return ZKHIP_OK;
    if ((size_t)256 * world > (size_t)zkhip_sc_tail_capacity()) return ZKHIP_OK;      // the gathered 256-entry tables must fit the tail
    const uint32_t g = log2_exact(cn) - 8;
    const uint32_t ov_k2 = overlapped_k2(cn), ov_k1 = g - ov_k2;
    BlockfoldShape sh;
    if (!blockfold_shape(1u << ov_k2, ov_k1, &sh)) return ZKHIP_ERR_SHAPE;
    *k1 = ov_k1; *k2 = ov_k2; *mid_entries = sh.ny << ov_k2;
    if (ny_out) *ny_out = sh.ny;
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_overlap_plan(zkhip_sc_state* st, uint32_t world, uint32_t* k1, uint32_t* k2, uint32_t* mid_entries) {
    if (!st || !k1 || !k2 || !mid_entries || !is_pow2(world)) return ZKHIP_ERR_ARG;
    *k1 = *k2 = *mid_entries = 0;
    if (st->round != 0 || st->stage_idx != 0 || st->ov_phase != 0) return ZKHIP_OK;
    uint32_t ny = 0;
    ZK_TRY(zk_sc_plan_overlap(st->cn, world, k1, k2, mid_entries, &ny));
    if (!*k1) return ZKHIP_OK;
    st->ov_k1 = *k1; st->ov_k2 = *k2; st->ov_ny1 = ny;
    st->stage_world = world;
    return ZKHIP_OK;
}
// d_out: the 2^k1 coarse block sums of the local shard (canonical integers; they add up across ranks all the same)
extern "C" int zkhip_sc_overlap_sums(zkhip_sc_state* st, uint64_t* d_out) {
    if (!st || !d_out || st->ov_k1 == 0 || st->ov_phase != 0) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    const uint32_t g = st->ov_k1 + st->ov_k2;
    ZK_TRY(launch_fine_sums(c, st->cur, st->cn, g, st->ov_fine(), nullptr));
    hipLaunchKernelGGL(group_sums_wg_kernel, dim3(1u << st->ov_k1), dim3(MLE_BLOCK), 0, c->stream, st->ov_fine(), 1u << st->ov_k2, (uint64_t*)nullptr, d_out);
    ZK_HIP(c, hipGetLastError());
    st->ov_phase = 1;
    return ZKHIP_OK;
}
// d_gathered: [world][2^k1] coarse sums in rank order.  Runs rounds 1..k1, starts the k1-variable fold of the shard on the fold
// stream and leaves this rank's partial block sums of the next k2 rounds in d_mid (mid_entries values).
extern "C" int zkhip_sc_overlap_rounds1(zkhip_sc_state* st, const uint64_t* d_gathered, uint32_t world, const uint64_t* h_claimed, uint64_t* d_mid) {
    if (!st || !d_gathered || !d_mid || st->ov_phase != 1 || world != st->stage_world) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    ZK_TRY(c->ensure_fold_stream());
    const uint32_t k1 = st->ov_k1, k2 = st->ov_k2;
    SmallArgs a = {};
    a.src = d_gathered; a.group = world; a.stride = 1u << k1; a.canon = 1; a.log_n = k1; a.n_rounds = k1; a.round0 = 0; a.first = 1;
    if (h_claimed) { std::memcpy(a.claimed.v, h_claimed, 32); a.first = 2; }
    a.weights_out = st->ov_w1(); a.final_out = nullptr;
    ZK_TRY(launch_small(c, a, st->dev(), st->rp(), st->ch()));
    uint32_t ny = 0, n_parts = 0;
    ZK_TRY(launch_blockfold(c, c->stream, st->ov_fine(), 1u << k2, k1, st->ov_w1(), d_mid, &ny));
    if (ny != st->ov_ny1) return ZKHIP_ERR_SHAPE;
    ZK_HIP(c, hipEventRecord(c->fork_ev, c->stream));
    ZK_HIP(c, hipStreamWaitEvent(c->fold_stream, c->fork_ev, 0));
    ZK_TRY(launch_multifold(c, c->fold_stream, st->cur, st->cn, k1, st->ov_w1(), st->A, st->spx(), &n_parts));
    ZK_HIP(c, hipEventRecord(c->join_ev, c->fold_stream));
    st->round = k1;
    st->ov_phase = 2;
    return ZKHIP_OK;
}
// d_gathered: [world][mid_entries].  Runs rounds k1+1..k1+k2 beside the fold, joins it and folds its output by the k2
// variables: the local table is 256 entries afterwards (zkhip_sc_local_table).
extern "C" int zkhip_sc_overlap_rounds2(zkhip_sc_state* st, const uint64_t* d_gathered, uint32_t world) {
    if (!st || !d_gathered || st->ov_phase != 2 || world != st->stage_world) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    const uint32_t k1 = st->ov_k1, k2 = st->ov_k2;
    SmallArgs b = {};
    b.src = d_gathered; b.group = world * st->ov_ny1; b.stride = 1u << k2; b.canon = 1; b.log_n = k2; b.n_rounds = k2; b.round0 = k1; b.first = 0;
    b.weights_out = st->ov_w2(); b.final_out = nullptr;
    // no exclusive-CU request here (cf. sumcheck_enqueue): the exchange puts this launch tens of microseconds behind the fold's
    // start, the chip is full by then and a whole free CU only appears when the fold drains
    ZK_TRY(launch_small(c, b, st->dev(), st->rp(), st->ch()));
    ZK_HIP(c, hipStreamWaitEvent(c->stream, c->join_ev, 0));
    uint32_t ny = 0;
    ZK_TRY(launch_blockfold(c, c->stream, st->A, 256, k2, st->ov_w2(), st->ov_p2(), &ny));
    if (ny > 32) return ZKHIP_ERR_SHAPE;
    hipLaunchKernelGGL(slice_sums_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, st->ov_p2(), ny, 256u, st->ov_loc());
    ZK_HIP(c, hipGetLastError());
    st->cur = st->ov_loc();
    st->cn = 256;
    st->round = k1 + k2;
    st->partials_valid = false;
    st->stage_idx++;
    st->ov_phase = 3;
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_local_len(zkhip_sc_state* st, size_t* n) {
    if (!st || !n) return ZKHIP_ERR_ARG;
    *n = st->cn;
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_local_half_sums(zkhip_sc_state* st, uint64_t* d_out) {
    if (!st || !d_out) return ZKHIP_ERR_ARG;
    if (st->cn < 2) return ZKHIP_ERR_SHAPE;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    if (!st->partials_valid) {   // first round (later rounds: the fold already produced them)
        const int grid = mle_grid((st->cn + 3) / 4);
        ProfScope ps(c, "half_sums", 32.0 * (double)st->cn);
        hipLaunchKernelGGL(half_sums_kernel, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, st->cur, st->cn, st->partials());
        st->np = (uint32_t)grid;
    }
    hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, st->partials(), st->np, st->fin());
    ZK_HIP(c, hipMemcpyAsync(d_out, st->fin(), 64, hipMemcpyDeviceToDevice, c->stream));
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_absorb(zkhip_sc_state* st, const uint64_t* d_gathered, uint32_t world, const uint64_t* h_claimed) {
    if (!st || !d_gathered || world == 0) return ZKHIP_ERR_ARG;
    if (st->round >= ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    FrArg claimed = {}, z = {};
    uint32_t first = st->round == 0 ? 1 : 0;
    if (first && h_claimed) { std::memcpy(claimed.v, h_claimed, 32); first = 2; }
    hipLaunchKernelGGL(sumcheck_round_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_gathered, world, st->dev(), st->round,
                       first, claimed, 0u, z, z, st->rp(), st->ch());
    ZK_HIP(c, hipGetLastError());
    st->round++;
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_fold(zkhip_sc_state* st) {
    if (!st) return ZKHIP_ERR_ARG;
    if (st->cn < 2 || st->round == 0) return ZKHIP_ERR_SHAPE;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    uint64_t* dst = (st->cur == st->A) ? st->B : st->A;
    const bool sums = st->cn >= 4;   // the folded table still has two halves to sum
    ZK_TRY(launch_fold(c, st->cur, st->cn, st->ch() + 4 * (st->round - 1), nullptr, 0, dst, sums, st->partials(), &st->np));
    st->partials_valid = sums;
    st->cur = dst;
    st->cn /= 2;
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_local_value(zkhip_sc_state* st, uint64_t* d_out) {
    if (!st || !d_out) return ZKHIP_ERR_ARG;
    if (st->cn != 1) return ZKHIP_ERR_SHAPE;
    ZK_HIP(st->c, hipMemcpyAsync(d_out, st->cur, 32, hipMemcpyDeviceToDevice, st->c->stream));
    return ZKHIP_OK;
}
extern "C" int zkhip_sc_local_table(zkhip_sc_state* st, uint64_t* d_out) {
    if (!st || !d_out) return ZKHIP_ERR_ARG;
    ZK_HIP(st->c, hipMemcpyAsync(d_out, st->cur, 32 * st->cn, hipMemcpyDeviceToDevice, st->c->stream));
    return ZKHIP_OK;
}
// The gathered table may be twice what the serial kernel holds: its first round then runs on its own (half sums, transcript,
// fold: four small launches) -- cheaper than one more round of the exchange protocol.
extern "C" int zkhip_sc_tail_capacity(void) { return 2 << TREE_MAX_LOG; }
extern "C" int zkhip_sc_tail(zkhip_sc_state* st, const uint64_t* d_values, uint32_t m, const uint64_t* h_claimed) {
    if (!st || !d_values) return ZKHIP_ERR_ARG;
    if (!is_pow2(m) || m > (2u << TREE_MAX_LOG)) return ZKHIP_ERR_SHAPE;
    if (m == 1) return ZKHIP_OK;
    zkhip_ctx* c = st->c;
    ZK_TRY(c->activate());
    uint32_t first = st->round == 0 ? 1u : 0u;
    FrArg claimed = {};
    if (first && h_claimed) { std::memcpy(claimed.v, h_claimed, 32); first = 2; }
    if (m > (1u << TREE_MAX_LOG)) {
        if (st->round >= ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
        FrArg z = {};
        const int grid = mle_grid((m + 3) / 4);
        uint64_t* folded = st->partials() + 64;              // m/2 entries behind the two records of half sums
        hipLaunchKernelGGL(half_sums_kernel, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, d_values, (size_t)m, st->partials());
        hipLaunchKernelGGL(sumcheck_round_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, st->partials(), (uint32_t)grid, st->dev(), st->round,
                           first, claimed, 0u, z, z, st->rp(), st->ch());
        ZK_TRY(launch_fold(c, d_values, m, st->ch() + 4 * st->round, nullptr, 0, folded, false, nullptr, nullptr));
        st->round++;
        first = 0;
        d_values = folded;
        m >>= 1;
    }
    SmallArgs a = {};
    a.src = d_values; a.group = 0; a.stride = 0; a.log_n = log2_exact(m); a.n_rounds = a.log_n; a.round0 = st->round;
    a.first = first; a.claimed = claimed;
    a.weights_out = nullptr; a.final_out = st->fin();
    ZK_TRY(launch_small(c, a, st->dev(), st->rp(), st->ch()));
    st->round += a.log_n;
    return ZKHIP_OK;
}
// releases a state's buffers and the context's loan flags
static void sc_release(zkhip_sc_state* st) {
    zkhip_ctx* c = st->c;
    if (st->A && st->owns_tables) hipFree(st->A);
    if (st->A && !st->owns_tables) c->ws_lent = false;
    if (st->uses_cache) {
        c->sc_lent = false;
    } else if (!c->sc_small) {           // keep this set for the next prove instead of freeing it
        c->sc_small = st->small; c->sc_stage = st->stage_buf; c->sc_stage_cap = st->stage_parts_cap;
    } else if (!c->sc_lent && st->stage_parts_cap > c->sc_stage_cap) {
        // the cached set is too small for shards of this size (a steady stream of them would allocate and free per prove): this larger
        // set replaces it.  (The caller has waited for the stream: finish / abort synchronise before they release.)
        hipFree(c->sc_small);
        if (c->sc_stage) hipFree(c->sc_stage);
        c->sc_small = st->small; c->sc_stage = st->stage_buf; c->sc_stage_cap = st->stage_parts_cap;
    } else {
        hipFree(st->small);
        if (st->stage_buf) hipFree(st->stage_buf);
    }
    delete st;
}
extern "C" int zkhip_sc_finish(zkhip_sc_state* st, uint64_t* h_sum, uint64_t* h_rp, uint64_t* h_ch, uint32_t* n_rounds) {
    if (!st) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = st->c;
    int rc = ZKHIP_OK;
    if (c->activate() != ZKHIP_OK) rc = ZKHIP_ERR_HIP;
    // one copy of [state | round polynomials | challenges] (contiguous in `small`) into pinned memory
    const size_t span = 64 + 12 * (size_t)ZK_MAX_ROUNDS;
    static_assert(ZK_PIN_END - ZK_PIN_PROOF >= 64 + 12 * ZK_MAX_ROUNDS, "pinned proof area too small");
    uint64_t* pin = c->pinned_u64(ZK_PIN_PROOF);
    if (rc == ZKHIP_OK && hipMemcpyAsync(pin, st->small, 8 * span, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    if (st->ov_phase == 2 && c->fold_stream && hipStreamSynchronize(c->fold_stream) != hipSuccess) rc = ZKHIP_ERR_HIP;   // un-joined fold (see zkhip_sc_abort)
    if (rc == ZKHIP_OK) {
        if (h_sum) std::memcpy(h_sum, pin + ((const uint64_t*)st->dev()->sum - st->small), 32);
        if (h_rp && st->round) std::memcpy(h_rp, pin + 64, 64 * (size_t)st->round);
        if (h_ch && st->round) std::memcpy(h_ch, pin + 64 + 8 * ZK_MAX_ROUNDS, 32 * (size_t)st->round);
    }
    if (n_rounds) *n_rounds = st->round;
    sc_release(st);
    return rc;
}
extern "C" int zkhip_sc_abort(zkhip_sc_state* st) {
    if (!st) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = st->c;
    int rc = ZKHIP_OK;
    if (c->activate() != ZKHIP_OK || hipStreamSynchronize(c->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;   // kernels may still read the buffers
    // between zkhip_sc_overlap_rounds1 and _rounds2 the shard's k1-variable fold runs on the fold stream and is joined to
    // c->stream only inside _rounds2: an abort in that window must wait for it too (it reads the caller's table and writes the
    // workspace and the cached stage buffer that sc_release hands back)
    if (c->fold_stream && hipStreamSynchronize(c->fold_stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    sc_release(st);
    return rc;
}

// ---- FiatShamirTranscript on bytes of the caller's choice (the device hash held against an independent SHA-256) ----
namespace zk {
static __global__ __launch_bounds__(64) void transcript_blocks_kernel(const uint32_t* __restrict__ blocks, uint32_t n_blocks, uint32_t* __restrict__ digest) {
    __shared__ uint32_t kw[64];
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    for (uint32_t b = 0; b < n_blocks; ++b) {
        uint32_t blk[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) blk[j] = blocks[16 * (size_t)b + j];
        sha256_compress_wave(h, blk, kw);
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) digest[i] = h[i];
    }
}
}  // namespace zk
extern "C" int zkhip_transcript_challenge(zkhip_ctx* c, const uint8_t* h_prefix32, const uint8_t* h_bytes, size_t n, uint8_t* h_digest32) {
    if (!c || !h_digest32 || (n && !h_bytes)) return ZKHIP_ERR_ARG;
    if (n >= ((size_t)1 << 31)) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    // the padded message as big-endian words (FIPS 180-4 5.1.1): prefix || bytes || 0x80 || 0 ... || bit length
    const size_t len = (h_prefix32 ? 32 : 0) + n, n_blocks = (len + 9 + 63) / 64;
    std::vector<uint8_t> msg(64 * n_blocks, 0);
    if (h_prefix32) std::memcpy(msg.data(), h_prefix32, 32);
    if (n) std::memcpy(msg.data() + (h_prefix32 ? 32 : 0), h_bytes, n);
    msg[len] = 0x80;
    const uint64_t bits = 8 * (uint64_t)len;
    for (int i = 0; i < 8; ++i) msg[64 * n_blocks - 1 - i] = (uint8_t)(bits >> (8 * i));
    std::vector<uint32_t> words(16 * n_blocks);
    for (size_t w = 0; w < words.size(); ++w)
        words[w] = ((uint32_t)msg[4 * w] << 24) | ((uint32_t)msg[4 * w + 1] << 16) | ((uint32_t)msg[4 * w + 2] << 8) | msg[4 * w + 3];
    ZK_TRY(c->reserve_ws(4 * words.size() + 64));
    uint32_t* d_words = (uint32_t*)c->d_ws;
    uint32_t* d_digest = d_words + words.size();
    ZK_HIP(c, hipMemcpyAsync(d_words, words.data(), 4 * words.size(), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(zk::transcript_blocks_kernel, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)d_words, (uint32_t)n_blocks, d_digest);
    ZK_HIP(c, hipGetLastError());
    uint32_t out[8];
    ZK_HIP(c, hipMemcpyAsync(out, d_digest, 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) h_digest32[4 * i + j] = (uint8_t)(out[i] >> (24 - 8 * j));
    return ZKHIP_OK;
}

#ifdef ZK_STAMPS
extern "C" int zkhip_debug_read_stamps(zkhip_ctx* c, unsigned long long* h_out /*64*8*/) {
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    ZK_HIP(c, hipMemcpyFromSymbol(h_out, HIP_SYMBOL(zk::g_zk_stamps), 64 * 8 * 8));
    return ZKHIP_OK;
}
#endif

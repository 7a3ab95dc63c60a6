// composed.hip -- C-ABI entry points of the composed / multi-composed sumcheck provers.
// gfx950 only.  No CPU fallback: every entry point launches HIP kernels or fails.
#include "../../include/zkhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ctx.hpp"
#include "host_util.hpp"
#include "composed_kernels.hpp"
#include "composed_stage.hpp"
#include "composed_dot.hpp"
#include "composed_pipe.hpp"
#include "host_fr.hpp"

using namespace zk;

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
    if (tp.lin_in) {   // term = product + additive table: instantiated for the shapes that occur (K <= 2, gkr.hip)
        if constexpr (K <= 2) {
            if (fold)
                hipLaunchKernelGGL((composed_round_kernel<K, true, true>), dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, n, r, rec, rec_off, partials);
            else
                hipLaunchKernelGGL((composed_round_kernel<K, false, true>), dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, n, r, rec, rec_off, partials);
        }
        return;
    }
    if constexpr (K == 5) {
        // large rounds: the last factor as a dot product on the matrix cores (composed_dot.hpp).  Measured (K = 5, one term, 512 workgroups,
        // profiles/r06/NOTES.md section 8): the first round 397 -> 293 us at 2^22 and 111 -> 92 us at 2^20; folding rounds 323 -> 303 us with
        // 2^20 output pairs, 172 -> 165 with 2^19, a draw with 2^18 -- so from 2^18 pairs on in the first round, 2^19 in the folding ones.
        // (K = 3, built and measured the same way: 7 -> 3 products in the first round, but its rounds at 2^20 are 35-55 us long and the staging,
        // the barriers and the closing reduction cost more than the products saved: 56 -> 67 us.  Not instantiated.)
        // ZKHIP_ROUND_DOT=0 keeps the vector form (A/B runs), ZKHIP_ROUND_DOT_MIN_LOG=<l> moves both thresholds (tests: the small sizes too).
        static const int dot_min_log = [] {
            const char* on = std::getenv("ZKHIP_ROUND_DOT");
            if (on && std::atoi(on) == 0) return 64;
            const char* e = std::getenv("ZKHIP_ROUND_DOT_MIN_LOG");
            return e ? -std::max(8, std::atoi(e)) : 18;
        }();
        const size_t work = fold ? n / 4 : n / 2;
        const int min_log = dot_min_log < 0 ? -dot_min_log : dot_min_log + (fold ? 1 : 0);
        const size_t per_wg = (work + (size_t)grid * CDT_ROWS - 1) / ((size_t)grid * CDT_ROWS) * CDT_ROWS;
        if (min_log < 64 && work >= ((size_t)1 << min_log) && per_wg <= CDT_MAX_PER_WG) {
            if (fold)
                hipLaunchKernelGGL((composed_round_dot_kernel<K, true>), dim3(grid), dim3(CDT_ROWS), cdt_lds_bytes(K), c->stream, tp, n, r, rec, rec_off, partials);
            else
                hipLaunchKernelGGL((composed_round_dot_kernel<K, false>), dim3(grid), dim3(CDT_ROWS), cdt_lds_bytes(K), c->stream, tp, n, r, rec, rec_off, partials);
            return;
        }
    }
    if (fold)
        hipLaunchKernelGGL((composed_round_kernel<K, true, false>), dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, n, r, rec, rec_off, partials);
    else
        hipLaunchKernelGGL((composed_round_kernel<K, false, false>), dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, n, r, rec, rec_off, partials);
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

// One composed / multi-composed sumcheck in progress: the state shared by the one-call provers below and by the
// split-phase session (zkhip_mc_*, tables sharded over several GPUs).  multi = 0: ComposedSumcheck (one term).
// lin_ptrs (nullable): per term an optional additive table (term = product + table; K <= 2).  cont: continue the
// transcript the previous call left in the context (its rounds are the next rounds of the same sumcheck) instead of
// starting one (the transcript lives in the context's ComposedDev, where every call's kernels leave it).
struct ComposedRun {
    zkhip_ctx* c = nullptr;
    ComposedMeta meta = {};
    uint32_t n_terms = 0, total = 0, total_all = 0, n_rounds = 0;
    uint32_t term_sizes[CMP_MAX_TERMS] = {};
    int multi = 0;
    size_t n = 0, cn = 0, per_table = 0;
    char* ws = nullptr;
    ComposedDev* st = nullptr;
    uint64_t *d_partials = nullptr, *d_rp = nullptr, *d_ch = nullptr;
    std::vector<const uint64_t*> cur, lin_cur;
    uint32_t round = 0, first = 1, tail_len = 0;
    uint32_t out_base = 0;   // rounds already recorded in d_rp / d_ch by an earlier call of the same sumcheck (cont): this call appends
    bool pending = false;    // the tables are still to be folded at the last challenge (the round kernels fold while they sum the next round)
    int cur_buf = 0;         // where the current tables lie: 0 the caller's, 1 the n/2-entry buffer, 2 the n/4-entry buffer
    uint64_t* d_stage_w = nullptr;   // fold weights of a two-round stage (composed_stage.hpp)
    uint64_t* d_pipe_rec[2] = {nullptr, nullptr};   // forms records of the pipelined rounds, alternating
    static constexpr size_t PIPE_MAX_WGS = 512;     // two per compute unit: all resident when the chip is this prover's alone
    static constexpr size_t PIPE_REC_BYTES = PIPE_MAX_WGS * 9 * CMP_MAX_TERMS * 32;
    uint32_t pipe_records = 0;       // > 0: the tail continues pipelined rounds -- forms records in d_pipe_rec[pipe_parity], tables unfolded
    int pipe_parity = 0;
    FrArg sum_arg = {};      // the claimed sum, passed to the closing kernels by value
    const uint64_t* sum_dev = nullptr;   // ... or read by them from device memory (gkr.hip: the kernel before computed it)
    OuterPub outer = {};     // an outer transcript fed beside the rounds (composed_kernels.hpp): every closing launch gets a hasher workgroup
    // RESIDENCY ASSUMPTION of every launch that adds extra_wg(): the hasher is the LAST workgroup of the grid and spin-waits (with a
    // time-out -> OuterDev::error -> ZKHIP_ERR_TIMEOUT) for a flag the closing workgroup of the SAME grid raises, so both must be resident
    // at once.  They are: the closer is dispatched first (workgroup 0 of a one- or two-workgroup grid; in composed_pipe_round_kernel the
    // closing workgroup comes behind at most PIPE_MAX_WGS = 2 x 256 tile workgroups that never wait for anything), and a grid of at most
    // 514 workgroups of <= 1024 lanes is less than the chip holds.  Other processes' kernels can delay the closer, not evict it.
    unsigned extra_wg() const { return outer.dev ? 1u : 0u; }

    // n = entries per table held here, n_rounds = rounds of the whole sumcheck (log2 n, more when other ranks hold shards)
    int setup(zkhip_ctx* ctx, const uint64_t* const* ptrs, const uint32_t* sizes, uint32_t nt, size_t n_entries, uint32_t rounds,
              int is_multi, const uint64_t* h_sum, int partial, const uint64_t* const* lin_ptrs, int cont) {
        c = ctx;
        n_terms = nt; n = cn = n_entries; n_rounds = rounds; multi = is_multi;
        meta.n_terms = n_terms;
        meta.multi = (uint32_t)multi;
        uint32_t rec = 0;
        total = 0;
        for (uint32_t p = 0; p < n_terms; ++p) {
            if (sizes[p] < 1 || sizes[p] > CMP_MAX_K) return ZKHIP_ERR_ARG;
            term_sizes[p] = meta.k[p] = sizes[p];
            meta.rec_off[p] = rec;
            rec += sizes[p] + 1;
            total += sizes[p];
        }
        meta.rec = rec;
        if (rec > CMP_MAX_REC) return ZKHIP_ERR_ARG;
        // additive tables: numbered after the product tables
        uint32_t n_lin = 0;
        lin_cur.assign(n_terms, nullptr);
        for (uint32_t p = 0; p < n_terms; ++p) {
            meta.lin_tab[p] = ~0u;
            if (lin_ptrs && lin_ptrs[p]) {
                if (sizes[p] > 2) return ZKHIP_ERR_ARG;
                lin_cur[p] = lin_ptrs[p];
                meta.lin_tab[p] = total + n_lin++;
            }
        }
        total_all = total + n_lin;
        tail_len = composed_tail_len(total_all);
        cur.assign(ptrs, ptrs + total);
        // workspace: per table a ping (n/2) and a pong (n/4) buffer, then the state
        per_table = (n / 2 + n / 4 + 2) * 32;
        const size_t w_off = (total_all * per_table + 255) & ~(size_t)255;
        const size_t pipe_off = w_off + 256;                         // two record buffers of the pipelined rounds (composed_pipe.hpp)
        const size_t bytes_off = pipe_off + 2 * PIPE_REC_BYTES;
        const size_t chunk = std::min<size_t>(n, (size_t)1 << 18);   // entries per staging buffer of prove()'s table-bytes pass
        ZK_TRY(c->reserve_ws(bytes_off + (multi && !partial ? 64 * chunk : 0)));
        ws = (char*)c->d_ws;
        d_stage_w = (uint64_t*)(ws + w_off);
        d_pipe_rec[0] = (uint64_t*)(ws + pipe_off);
        d_pipe_rec[1] = (uint64_t*)(ws + pipe_off + PIPE_REC_BYTES);
        pending = false;
        cur_buf = 0;
        // the device-resident state (transcript, interpolation matrices of every degree) belongs to the context: uploaded once,
        // and a continuation (cont) finds the transcript where the previous call's kernels left it -- no copies, no waiting
        if (!c->d_composed) {
            std::vector<uint64_t> img(sizeof(ComposedDev) / 8, 0);
            uint64_t* mats = img.data() + offsetof(ComposedDev, interp) / 8;
            for (int d = 1; d <= CMP_MAX_K; ++d) {
                std::vector<zkhost::Fr> m = zkhost::interpolation_matrix(d);
                std::memcpy(&mats[(size_t)d * (CMP_MAX_K + 1) * (CMP_MAX_K + 1) * 4], m.data(), m.size() * 32);
            }
            void* mem = nullptr;
            if (hipMalloc(&mem, sizeof(ComposedDev)) != hipSuccess) return ZKHIP_ERR_NOMEM;
            if (hipMemcpy(mem, img.data(), sizeof(ComposedDev), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(mem); return ZKHIP_ERR_HIP; }
            c->d_composed = mem;
        }
        st = (ComposedDev*)c->d_composed;
        d_partials = c->small_u64(ZK_SMALL_PARTIALS);
        d_rp = c->small_u64(ZK_SMALL_ROUNDPOLYS);
        d_ch = c->small_u64(ZK_SMALL_CHALLENGES);
        first = 1;
        round = 0;
        if (cont) first = 0;                                   // st->transcript is the previous call's
        if (!multi) return ZKHIP_OK;
        if (h_sum) std::memcpy(sum_arg.v, h_sum, 32);
        if (!partial && !cont) {
            // prove(): transcript.commit(&composed_poly_to_bytes(&poly)) first (multi_composed_sumcheck.rs:51-53).
            // The GPU produces the canonical big-endian bytes, the host hashes the (inherently sequential) stream.
            // Chunks of 2^18 entries (8 MiB of bytes) through two device staging buffers and two pinned buffers: the
            // conversion + copy of chunk k + 1 runs while the host hashes chunk k (SHA extensions where the CPU has them).
            uint8_t* d_bytes = (uint8_t*)(ws + bytes_off);
            const size_t per_table = (n + chunk - 1) / chunk, n_chunks = per_table * total;
            ZK_TRY(c->reserve_msm_pin(0, 32 * chunk));
            ZK_TRY(c->reserve_msm_pin(1, 32 * chunk));
            zkhost::Sha256 sha;
            auto chunk_len = [&](size_t k) { const size_t off = (k % per_table) * chunk; return std::min(chunk, n - off); };
            auto issue = [&](size_t k) -> int {
                const size_t q = k / per_table, off = (k % per_table) * chunk, len = chunk_len(k);
                const int sl = (int)(k & 1);
                uint8_t* stage = d_bytes + (size_t)sl * 32 * chunk;
                hipLaunchKernelGGL(to_bytes_kernel, dim3(mle_grid_stream(len)), dim3(MLE_BLOCK), 0, c->stream, ptrs[q] + 4 * off, len, (uint32_t*)stage);
                ZK_HIP(c, hipMemcpyAsync(c->msm_pin[sl], stage, 32 * len, hipMemcpyDeviceToHost, c->stream));
                ZK_HIP(c, hipEventRecord(c->msm_ev[sl], c->stream));
                return ZKHIP_OK;
            };
            ZK_TRY(issue(0));
            for (size_t k = 0; k < n_chunks; ++k) {
                if (k + 1 < n_chunks) ZK_TRY(issue(k + 1));            // its buffers held chunk k - 1, hashed in the last iteration
                ZK_HIP(c, hipEventSynchronize(c->msm_ev[k & 1]));
                sha.update((const uint8_t*)c->msm_pin[k & 1], 32 * chunk_len(k));
            }
            Sha256State hs = {};
            std::memcpy(hs.h, sha.h, 32);
            const uint32_t fill = (uint32_t)(sha.len % 64);
            for (uint32_t i = 0; i < fill / 4; ++i)
                hs.buf[i] = ((uint32_t)sha.buf[4 * i] << 24) | ((uint32_t)sha.buf[4 * i + 1] << 16) | ((uint32_t)sha.buf[4 * i + 2] << 8) | sha.buf[4 * i + 3];
            hs.fill = fill;
            hs.len = sha.len;
            ZK_HIP(c, hipMemcpyAsync(&st->transcript, &hs, sizeof(hs), hipMemcpyHostToDevice, c->stream));
            ZK_HIP(c, hipStreamSynchronize(c->stream));   // hs is a stack temporary
            first = 2;
        }
        return ZKHIP_OK;
    }
    bool folds() const { return pending; }
    const uint64_t* prev_challenge() const { return d_ch + 4 * (size_t)(out_base + round - 1); }   // valid when folds()
    size_t after() const { return folds() ? cn / 2 : cn; }   // entries the current round's sums run over
    CloseArgs close_args() const {
        CloseArgs ca = {};
        ca.meta = meta; ca.st = st; ca.round = out_base + round; ca.first = first; ca.round_out = d_rp; ca.challenges = d_ch; ca.sum = sum_arg;
        ca.sum_dev = sum_dev; ca.outer = outer;
        return ca;
    }
    // The round on tables too large for one workgroup's LDS, first part: one launch per term (fold at the previous
    // challenge + the round's sums, one record per workgroup in d_partials).  Returns the number of records.
    int round_sums(int* n_records) {
        const bool fold = folds();
        const size_t work = fold ? cn / 4 : cn / 2;
        // small folding rounds of K = 2 terms: four lanes per output pair (composed_round_split2_kernel), all terms in one launch
        bool split = fold && work >= 1 && work <= CMP_SPLIT_MAX;
        for (uint32_t p = 0; p < n_terms; ++p) split = split && term_sizes[p] == 2;
        // the first rounds of a LARGE claim of one K = 2 term: unreduced sums of products, >= 8 (folding) / 16 pairs per lane
        // (composed_round_wide2_kernel); at most 256 products per accumulator
        const size_t per_lane = fold ? 8 : 16;
        const bool wide = n_terms == 1 && term_sizes[0] == 2 && !lin_cur[0] && work >= CMP_WIDE_MIN_WORK &&
                          work <= (size_t)MLE_MAX_GRID * MLE_BLOCK * 256;
        // one record per workgroup, and the closing kernel -- one workgroup, on the critical path -- adds them up: a claim of ONE term
        // launches at most 512 workgroups (K = 2 at 2^22: 0.626 / 0.621 / 0.614-0.621 / 0.619 ms at 2048 / 1024 / 512 / 256; the GKR layers,
        // two terms per claim, measured no better with fewer).  ZKHIP_ROUND_GRID overrides (diagnostics).
        static const int grid_env = [] { const char* e = std::getenv("ZKHIP_ROUND_GRID"); return e ? std::atoi(e) : 0; }();
        const int grid_cap = grid_env > 0 ? grid_env : n_terms == 1 ? 512 : (int)MLE_MAX_GRID;
        // small folding rounds with a term of three and more tables: K + 1 lanes per output pair (composed_round_tsplit_kernel), ONE pass
        // per workgroup -- the grid is sized for the widest such term and every term of the round uses it (the records of a round
        // are per workgroup).  ZKHIP_ROUND_TSPLIT=0: off (A/B runs).
        static const bool tsplit_on = [] { const char* e = std::getenv("ZKHIP_ROUND_TSPLIT"); return !e || std::atoi(e) != 0; }();
        uint32_t k_wide = 0;
        bool any_lin = false;
        for (uint32_t p = 0; p < n_terms; ++p) { k_wide = std::max(k_wide, term_sizes[p]); any_lin = any_lin || lin_cur[p] != nullptr; }
        const bool tsplit = tsplit_on && fold && k_wide >= 3 && !any_lin && work >= 1 && work <= CMP_TSPLIT_MAX;
        const size_t tsplit_per_wg = (size_t)(MLE_BLOCK / 64) * (64 / (k_wide + 1));
        const int grid = tsplit ? (int)std::min<size_t>(MLE_MAX_GRID, (work + tsplit_per_wg - 1) / tsplit_per_wg)
                         : split ? (int)((4 * work + MLE_BLOCK - 1) / MLE_BLOCK)
                         : wide ? (int)std::min<size_t>(MLE_MAX_GRID, std::max<size_t>(256, work / (MLE_BLOCK * per_lane)))
                         : std::min(grid_cap, mle_grid(work ? work : 1));
        uint32_t off = 0;
        // every term with the same number of tables (<= 2 when one has an additive table): ONE launch, blockIdx.y = term
        bool same_k = n_terms > 1;
        for (uint32_t p = 1; p < n_terms; ++p) same_k = same_k && term_sizes[p] == term_sizes[0];
        for (uint32_t p = 0; p < n_terms; ++p) if (lin_cur[p] && term_sizes[p] > 2) same_k = false;
        if (tsplit) same_k = false;     // K + 1 lanes per pair, every term in one launch (composed_round_tsplit_kernel) below
        same_k = same_k || split;
        MultiTablePtrs mp = {};
        TsplitTerms ts = {};
        for (uint32_t p = 0; p < n_terms; ++p) {
            TablePtrs tp = {};
            // ping-pong: a fold writes into the buffer the current tables do not lie in (out_buf)
            for (uint32_t q = 0; q < term_sizes[p]; ++q) {
                tp.in[q] = cur[off + q];
                tp.out[q] = out_buf(off + q);
            }
            if (lin_cur[p]) {
                tp.lin_in = lin_cur[p];
                tp.lin_out = out_buf(meta.lin_tab[p]);
            }
            if (tsplit) {
                ts.t[p] = tp;
                ts.rec_off[p] = meta.rec_off[p];
                ts.k[p] = term_sizes[p];
            } else if (same_k) {
                mp.t[p] = tp;
                mp.rec_off[p] = meta.rec_off[p];
            } else {
                ProfScope ps(c, "composed_round", 0.0);
                if (wide) {
                    if (fold) hipLaunchKernelGGL(composed_round_wide2_kernel<true>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, cn, prev_challenge(), meta.rec, meta.rec_off[p], d_partials);
                    else hipLaunchKernelGGL(composed_round_wide2_kernel<false>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, tp, cn, (const uint64_t*)nullptr, meta.rec, meta.rec_off[p], d_partials);
                } else {
#define CALL(KK) launch_round<KK>(c, fold, tp, cn, fold ? prev_challenge() : nullptr, meta.rec, meta.rec_off[p], d_partials, grid)
                ZK_DISPATCH_K(term_sizes[p], CALL)
#undef CALL
                }
            }
            if (fold) for (uint32_t q = 0; q < term_sizes[p]; ++q) cur[off + q] = tp.out[q];
            if (fold && lin_cur[p]) lin_cur[p] = tp.lin_out;
            off += term_sizes[p];
        }
        if (tsplit) {
            ProfScope ps(c, "composed_round", 0.0);
            hipLaunchKernelGGL(composed_round_tsplit_kernel, dim3(grid, n_terms), dim3(MLE_BLOCK), 0, c->stream, ts, cn, prev_challenge(), meta.rec, d_partials);
        }
        if (same_k) {
            ProfScope ps(c, "composed_round", 0.0);
            const uint64_t* rp = fold ? prev_challenge() : nullptr;
            if (split) hipLaunchKernelGGL(composed_round_split2_kernel, dim3(grid, n_terms), dim3(MLE_BLOCK), 0, c->stream, mp, cn, rp, meta.rec, d_partials);
            else {
#define CALL(KK)                                                                                                                                   \
            if (fold) hipLaunchKernelGGL((composed_round_multi_kernel<KK, true>), dim3(grid, n_terms), dim3(MLE_BLOCK), 0, c->stream, mp, cn, rp, meta.rec, d_partials); \
            else hipLaunchKernelGGL((composed_round_multi_kernel<KK, false>), dim3(grid, n_terms), dim3(MLE_BLOCK), 0, c->stream, mp, cn, rp, meta.rec, d_partials)
            ZK_DISPATCH_K(term_sizes[0], CALL)
#undef CALL
            }
        }
        if (fold) { cn /= 2; cur_buf = cur_buf == 1 ? 2 : 1; }
        *n_records = grid;
        return ZKHIP_OK;
    }
    // the buffer a fold of table slot `q` writes: the n/2-entry one unless the current tables lie there (then they hold <= n/2 entries
    // and the result fits the n/4-entry one)
    uint64_t* out_buf(uint32_t q) const {
        char* base = ws + (size_t)q * per_table;
        return (uint64_t*)(cur_buf == 1 ? base + (n / 2 + 1) * 32 : base);
    }
    // ---- two rounds per pass (composed_stage.hpp): every term a product of two tables, nothing pending, and enough entries that the
    // passes are throughput work (below that the four-lane round kernel and the LDS tail are faster)
    // Single GPU: a stage is cross sums + ONE serial kernel (two transcript rounds, ~36 us) + a fold by both challenges against two fused
    // round kernels + two closing kernels (~13 us each).  With the cross sums on the VALU (16 products per index against 13 for two
    // rounds) it never paid; with them on the matrix cores (composed_cross2_mfma_kernel: 53 us at 2^22 against 113) it pays where the
    // passes are long: measured per stage / per two rounds (tools/prof_composed_k2.py, one term) 156 / 195 us at 2^22, 88 / 81 at 2^20,
    // 62 / 46 at 2^18; claims of several terms (no unreduced wide kernel in their round form) gain from 2^18 on.  With the serial kernel's
    // second round one round ahead of the transcript (composed_stage_close_pipe_kernel, ~18 us less per stage) one-term claims gain from
    // 2^18 on as well: ComposedSumcheck at 2^22 0.528 / 0.525 / 0.521 / 0.518 / 0.520 ms with stages from 2^21 / 2^20 / 2^19 / 2^18 / 2^17
    // (tools/sweep_stage.sh, same box).  ZKHIP_STAGE=0 / 1 forces it off / on from 2^15 (measurements); the sharded sessions use it
    // wherever their shard allows: there it saves an EXCHANGE.
    bool stage_possible(size_t min_n) const {
        if (pending || cn < min_n || cn < 4 || n_rounds - round < 2) return false;
        for (uint32_t p = 0; p < n_terms; ++p) if (term_sizes[p] != 2) return false;
        return true;
    }
    bool stage_ok() const {
        static const int mode = [] { const char* e = std::getenv("ZKHIP_STAGE"); return e ? std::atoi(e) : -1; }();
        if (mode == 0) return false;
        if (mode == 1) return stage_possible((size_t)1 << 15);
        static const int log1 = [] { const char* e = std::getenv("ZKHIP_STAGE_MIN_LOG_ONE"); const int v = e ? std::atoi(e) : 0; return v >= 12 && v <= 30 ? v : 18; }();     // tuning aids (tools/sweep_stage.sh)
        static const int logm = [] { const char* e = std::getenv("ZKHIP_STAGE_MIN_LOG_MANY"); const int v = e ? std::atoi(e) : 0; return v >= 12 && v <= 30 ? v : 18; }();
        return stage_possible(n_terms >= 2 ? (size_t)1 << logm : (size_t)1 << log1);
    }
    MultiTablePtrs stage_tables() const {
        MultiTablePtrs mp = {};
        uint32_t off = 0;
        for (uint32_t p = 0; p < n_terms; ++p) {
            mp.t[p].in[0] = cur[off];
            mp.t[p].in[1] = cur[off + 1];
            mp.t[p].lin_in = lin_cur[p];
            mp.rec_off[p] = meta.rec_off[p];
            off += 2;
        }
        return mp;
    }
    // first part: the cross-block sums, one record of n_terms * CST_VALS values per workgroup in d_partials
    int stage_sums(int* n_records) {
        const size_t m = cn / 4;
        ProfScope ps(c, "composed_cross2", 0.0);
        static const bool no_mfma = [] { const char* e = std::getenv("ZKHIP_CROSS_VALU"); return e && std::atoi(e) != 0; }();   // diagnostics: A/B
        if (!no_mfma && m >= 1024 && m <= (size_t)256 * 65536) {
            // byte outer products on the matrix cores: >= 128 indices and <= 65536 per workgroup (int32 accumulators)
            // up to four workgroups per CU (42 KiB of LDS each), as many records as the scratch holds
            static const size_t cap_env = [] { const char* e = std::getenv("ZKHIP_CROSS_GRID"); return e ? (size_t)std::atoi(e) : (size_t)0; }();   // diagnostics
            const size_t cap = std::min<size_t>(cap_env ? cap_env : 512, (size_t)(8 * ZK_MAX_PARTIALS) / ((size_t)CST_VALS * n_terms));
            const int grid = (int)std::max<size_t>(std::min<size_t>(cap, m / 128), (m + 65535) / 65536);
            hipLaunchKernelGGL(composed_cross2_mfma_kernel, dim3(grid, n_terms), dim3(256), 0, c->stream, stage_tables(), cn, n_terms, d_partials);
            *n_records = grid;
            return ZKHIP_OK;
        }
        const int grid = (int)std::min<size_t>(CST_MAX_GRID, std::max<size_t>(1, m / 256));
        hipLaunchKernelGGL(composed_cross2_kernel, dim3(grid, n_terms), dim3(CST_CROSS_BLOCK), 0, c->stream, stage_tables(), cn, n_terms, d_partials);
        *n_records = grid;
        return ZKHIP_OK;
    }
    // second part: sum the records (the workgroups' here, the ranks' in the sharded protocol), run the two rounds, fold every table by both
    int stage_close(const uint64_t* records, uint32_t n_records) {
        StageArgs sa = {};
        sa.ca = close_args();
        sa.n_records = n_records;
        sa.weights_out = d_stage_w;
        {
            ProfScope ps(c, "composed_stage_close", 0.0);
            // the second round one round ahead of the transcript (composed_pipe.hpp); ZKHIP_PIPE=0 keeps the round-by-round form (same-box A/B)
            if (pipe_on() && pipe_eligible(meta))
                hipLaunchKernelGGL(composed_stage_close_pipe_kernel, dim3(1 + extra_wg()), dim3(PIPE_BLOCK), 0, c->stream, records, sa);
            else
                hipLaunchKernelGGL(composed_stage_close_kernel, dim3(1 + extra_wg()), dim3(CST_BLOCK), 0, c->stream, records, sa);
        }
        first = 0;
        round += 2;
        const size_t m = cn / 4;
        Fold2Tables ft = {};
        uint32_t nt = 0;
        for (uint32_t q = 0; q < total; ++q) { ft.in[nt] = cur[q]; ft.out[nt] = out_buf(q); cur[q] = ft.out[nt]; ++nt; }
        for (uint32_t p = 0; p < n_terms; ++p)
            if (lin_cur[p]) { ft.in[nt] = lin_cur[p]; ft.out[nt] = out_buf(meta.lin_tab[p]); lin_cur[p] = ft.out[nt]; ++nt; }
        {
            ProfScope ps(c, "composed_fold2", 0.0);
            const int grid = (int)std::min<size_t>(2048, (m + CST_BLOCK - 1) / CST_BLOCK);
            hipLaunchKernelGGL(composed_fold2_kernel, dim3(grid, nt), dim3(CST_BLOCK), 0, c->stream, ft, m, (const uint64_t*)d_stage_w);
        }
        cn = m;
        cur_buf = cur_buf == 1 ? 2 : 1;
        return ZKHIP_OK;
    }
    int stage() {
        int grid = 0;
        ZK_TRY(stage_sums(&grid));
        return stage_close(d_partials, (uint32_t)grid);
    }
    // second part: sum the records (the workgroups' here, the ranks' in the sharded protocol) and close the round
    void close(const uint64_t* records, uint32_t n_records) {
        hipLaunchKernelGGL(composed_close_kernel, dim3(1 + extra_wg()), dim3(MLE_BLOCK), 0, c->stream, records, n_records, close_args());
        first = 0;
        ++round;
        pending = true;
    }
    // all remaining rounds in one launch, on tables of m = after() entries that fit the LDS (fold: they are still to be
    // folded at the previous challenge while loading)
    int tail(const TailTables& tt, uint32_t m, bool fold) {
        ProfScope ps(c, "composed_tail", 0.0);
        // claims whose terms are products of two tables (every GKR layer, the reference's composed bench shape): one round AHEAD of the
        // transcript (composed_pipe.hpp).  ZKHIP_PIPE=0 keeps the round-by-round tail (same-box A/B).
        if (pipe_on() && pipe_eligible(meta)) {
            ZK_TRY(c->allow_big_lds((const void*)composed_tail_pipe_kernel, (size_t)CMP_TAIL_ENTRIES * 32));
            hipLaunchKernelGGL(composed_tail_pipe_kernel, dim3(1 + extra_wg()), dim3(PIPE_BLOCK), (size_t)total_all * m * 32, c->stream, tt, total_all, m,
                               fold ? 1u : 0u, fold ? prev_challenge() : nullptr, close_args(), n_rounds - round, (const uint64_t*)nullptr, 0u, pipe_max_q());
        } else {
            ZK_TRY(c->allow_big_lds((const void*)composed_tail_kernel, (size_t)CMP_TAIL_ENTRIES * 32));
            hipLaunchKernelGGL(composed_tail_kernel, dim3(1 + extra_wg()), dim3(CMP_TAIL_BLOCK), (size_t)total_all * m * 32, c->stream, tt, total_all, m,
                               fold ? 1u : 0u, fold ? prev_challenge() : nullptr, close_args(), n_rounds - round);
        }
        round = n_rounds;
        return ZKHIP_OK;
    }
    static uint32_t pipe_max_q() {     // forms are computed ahead for tables of <= 4 max_q entries (ZKHIP_PIPE_MAX_Q: tuning)
        static const uint32_t v = [] { const char* e = std::getenv("ZKHIP_PIPE_MAX_Q"); return e ? (uint32_t)std::atoi(e) : PIPE_MAX_Q; }();
        return v;
    }
    static size_t pipe_tail_max() {    // tables of at most this many entries go to the single-workgroup tail behind pipelined rounds (ZKHIP_PIPE_TAIL: tuning)
        static const size_t v = [] { const char* e = std::getenv("ZKHIP_PIPE_TAIL"); return e ? (size_t)std::atoi(e) : (size_t)512; }();
        return v;
    }
    static bool pipe_on() {
        static const bool on = [] { const char* e = std::getenv("ZKHIP_PIPE"); return !e || std::atoi(e) != 0; }();
        return on;
    }
    // ---- the rounds between the streaming sizes and the LDS tail, one launch per round and one round ahead (composed_pipe.hpp) ----
    // Entered with tables of cn entries (a fold at the last challenge pending or not) that do not fit the tail yet.
    static constexpr size_t PIPE_MID_MAX = (size_t)1 << 17;      // entries per table: above, the streaming forms (stages, wide rounds) are faster
    bool pipe_mid_ok() const {
        static const bool mid_on = [] { const char* e = std::getenv("ZKHIP_PIPE_MID"); return !e || std::atoi(e) != 0; }();
        return pipe_on() && mid_on && pipe_eligible(meta) && after() > tail_len && after() <= PIPE_MID_MAX && after() >= 4 * PIPE_TILE && n_rounds - round >= 2;
    }
    MultiTablePtrs pipe_tables(bool with_out) const {
        MultiTablePtrs mp = {};
        uint32_t off = 0;
        for (uint32_t p = 0; p < n_terms; ++p) {
            for (uint32_t q = 0; q < 2; ++q) { mp.t[p].in[q] = cur[off + q]; mp.t[p].out[q] = with_out ? out_buf(off + q) : nullptr; }
            mp.t[p].lin_in = lin_cur[p];
            mp.t[p].lin_out = (with_out && lin_cur[p]) ? out_buf(meta.lin_tab[p]) : nullptr;
            mp.rec_off[p] = meta.rec_off[p];
            off += 2;
        }
        return mp;
    }
    int pipe_launch(bool fold, uint32_t do_close, const uint64_t* rec_in, uint32_t n_rec_in, uint64_t* rec_out, uint32_t* n_rec_out) {
        PipeRoundArgs a = {};
        a.ca = close_args();
        a.tabs = pipe_tables(fold);
        a.cn = fold ? cn / 2 : cn;
        a.fold = fold ? 1u : 0u;
        a.fold_round = out_base + round - 1;
        a.do_close = do_close;
        a.records_in = rec_in; a.n_records_in = n_rec_in; a.records_out = rec_out;
        const size_t tiles = std::max<size_t>(1, (a.cn / 4 + PIPE_TILE - 1) / PIPE_TILE);
        const uint32_t n_cross = (uint32_t)std::min<size_t>(tiles, pipe_wgs());
        const size_t lds = (size_t)3 * n_terms * 4 * PIPE_TILE * 32;
        ZK_TRY(c->allow_big_lds((const void*)composed_pipe_round_kernel, 128 * 1024));
        ProfScope ps(c, "composed_pipe_round", 0.0);
        hipLaunchKernelGGL(composed_pipe_round_kernel, dim3(n_cross + (do_close ? 1 + extra_wg() : 0)), dim3(PIPE_BLOCK), lds, c->stream, a);
        *n_rec_out = n_cross;
        return ZKHIP_OK;
    }
    // the steady rounds of pipe_mid from the current state on: how many there are
    uint32_t pipe_steady_rounds() const {
        size_t cn_ = cn;
        uint32_t round_ = round, k = 0;
        while ((cn_ > tail_len || cn_ > pipe_tail_max()) && cn_ >= 8 * PIPE_TILE && n_rounds - round_ >= 2) { cn_ /= 2; ++round_; ++k; }
        return k;
    }
    static size_t pipe_wgs() {         // workgroups that take tiles, at most (ZKHIP_PIPE_WGS: tuning)
        static const size_t v = [] { const char* e = std::getenv("ZKHIP_PIPE_WGS"); const int x = e ? std::atoi(e) : 0; return x >= 1 && x <= (int)PIPE_MAX_WGS ? (size_t)x : (size_t)256; }();
        return v;
    }
    // one steady round's bookkeeping (what a launch of composed_pipe_round_kernel with fold = 1 leaves behind)
    void pipe_advance() {
        for (uint32_t q = 0; q < total; ++q) cur[q] = out_buf(q);
        for (uint32_t p = 0; p < n_terms; ++p) if (lin_cur[p]) lin_cur[p] = out_buf(meta.lin_tab[p]);
        cn /= 2; cur_buf = cur_buf == 1 ? 2 : 1;
        ++round;
        pipe_parity ^= 1;
    }
    int pipe_mid() {
        // the first round the round-by-round way, with the forms of the NEXT round computed between its two launches
        int grid = 0;
        ZK_TRY(round_sums(&grid));                       // (folds at the previous challenge on the way, if one was pending)
        uint32_t n_rec = 0;
        pipe_parity = 0;
        // ONE launch closes that round from the sums and computes the next round's forms beside it
        ZK_TRY(pipe_launch(false, 2u, d_partials, (uint32_t)grid, d_pipe_rec[0], &n_rec));
        first = 0;
        ++round;
        pending = true;                                  // a fold at the new challenge
        // steady state: one launch closes a round and prepares the next one
        const uint32_t k = pipe_steady_rounds();
        for (uint32_t i = 0; i < k; ++i) {
            uint32_t n_out = 0;
            ZK_TRY(pipe_launch(true, 1u, d_pipe_rec[pipe_parity], n_rec, d_pipe_rec[pipe_parity ^ 1], &n_out));
            pipe_advance();
            n_rec = n_out;
        }
        pipe_records = n_rec;
        return ZKHIP_OK;
    }
    // the tail behind pipelined rounds: the tables of the last closed round (cn entries, unfolded) + the forms of the next one
    int tail_after_pipe() {
        ZK_TRY(c->allow_big_lds((const void*)composed_tail_pipe_kernel, (size_t)CMP_TAIL_ENTRIES * 32));
        ProfScope ps(c, "composed_tail", 0.0);
        hipLaunchKernelGGL(composed_tail_pipe_kernel, dim3(1 + extra_wg()), dim3(PIPE_BLOCK), (size_t)total_all * cn * 32, c->stream, current_tables(), total_all,
                           (uint32_t)cn, 0u, (const uint64_t*)nullptr, close_args(), n_rounds - round, (const uint64_t*)d_pipe_rec[pipe_parity], pipe_records, pipe_max_q());
        pipe_records = 0;
        round = n_rounds;
        return ZKHIP_OK;
    }
    TailTables current_tables() const {
        TailTables tt = {};
        for (uint32_t q = 0; q < total; ++q) tt.in[q] = cur[q];
        for (uint32_t p = 0; p < n_terms; ++p) if (lin_cur[p]) tt.in[meta.lin_tab[p]] = lin_cur[p];
        return tt;
    }
    // round polynomials and challenges to the host, in the layouts of the C ABI
    // launch errors of everything enqueued so far (the transcript already lives in the context's persistent state, where a
    // continuation picks it up)
    int check_launches() {
        ZK_HIP(c, hipGetLastError());
        return ZKHIP_OK;
    }
    int collect(uint32_t* h_lens, uint64_t* h_round_polys, uint64_t* h_challenges) {
        ZK_TRY(check_launches());
        return collect_rounds(c, multi, term_sizes[0], out_base + n_rounds, h_lens, h_round_polys, h_challenges);
    }
    // the first `rounds` recorded rounds (of this call and the calls it continued) to the host
    static int collect_rounds(zkhip_ctx* c, int multi, uint32_t k0, uint32_t n_rounds, uint32_t* h_lens, uint64_t* h_round_polys,
                              uint64_t* h_challenges) {
        // ONE copy into pinned memory: the challenges and the recorded rounds are neighbours in the context's small buffer.  (Two copies into
        // the caller's pageable buffers were two staged, host-blocking transfers behind the last kernel of every proof and GKR layer.)
        static_assert(ZK_SMALL_ROUNDPOLYS == ZK_SMALL_CHALLENGES + 4 * ZK_MAX_ROUNDS, "challenges | round polynomials, adjacent");
        static_assert(ZK_PIN_END - ZK_PIN_PROOF >= 4 * ZK_MAX_ROUNDS + 64 * ZK_MAX_ROUNDS, "pinned proof area too small");
        if (n_rounds > (uint32_t)ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
        const uint64_t* d_ch = c->small_u64(ZK_SMALL_CHALLENGES);
        const uint32_t term_sizes[1] = {k0};
        uint64_t* pin = c->pinned_u64(ZK_PIN_PROOF);
        ZK_HIP(c, hipMemcpyAsync(pin, d_ch, 8 * (4 * (size_t)ZK_MAX_ROUNDS + 64 * (size_t)n_rounds), hipMemcpyDeviceToHost, c->stream));
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        std::memcpy(h_challenges, pin, 32 * (size_t)n_rounds);
        const uint64_t* h_rp = pin + 4 * ZK_MAX_ROUNDS;
        for (uint32_t r = 0; r < n_rounds; ++r) {
            if (!multi) {
                std::memcpy(h_round_polys + (size_t)r * (term_sizes[0] + 1) * 4, &h_rp[64 * r], (term_sizes[0] + 1) * 32);
            } else {
                h_lens[r] = (uint32_t)h_rp[64 * r];
                std::memcpy(h_round_polys + (size_t)r * CMP_MAX_MONO * 8, &h_rp[64 * r + 8], CMP_MAX_MONO * 64);
            }
        }
        return ZKHIP_OK;
    }
};

static int composed_prove_impl(zkhip_ctx* c, const uint64_t* const* ptrs, const uint32_t* term_sizes, uint32_t n_terms,
                               size_t n, int multi, const uint64_t* h_sum, int partial, uint32_t* h_lens,
                               uint64_t* h_round_polys, uint64_t* h_challenges, const uint64_t* const* lin_ptrs = nullptr,
                               int cont = 0) {
    if (!c || !ptrs || !term_sizes) return ZKHIP_ERR_ARG;
    if (n_terms == 0 || n_terms > CMP_MAX_TERMS) return ZKHIP_ERR_ARG;
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;
    const uint32_t n_vars = log2_exact(n);
    if (n_vars > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    if (n_vars == 0) return ZKHIP_OK;   // `for _ in 0..n_vars` never runs
    if (!h_round_polys || !h_challenges || (multi && ((!h_sum && !cont) || !h_lens))) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    ComposedRun run;
    ZK_TRY(run.setup(c, ptrs, term_sizes, n_terms, n, n_vars, multi, h_sum, partial, lin_ptrs, cont));
    // From the round whose tables fit the LDS on, one launch finishes the proof.
    while (run.round < n_vars) {
        if (run.after() <= run.tail_len) {
            ZK_TRY(run.tail(run.current_tables(), (uint32_t)run.after(), run.folds()));
            break;
        }
        if (run.stage_ok()) { ZK_TRY(run.stage()); continue; }
        if (run.pipe_mid_ok()) {                 // one launch per round, one round ahead, down to the LDS tail
            ZK_TRY(run.pipe_mid());
            // pipe_mid() ends with the tables still unfolded at the last closed round's challenge and that challenge's Montgomery form
            // not yet in d_ch: only tail_after_pipe() continues from there.  It runs down to the LDS tail by construction (tail_len >=
            // 256 >= the steady loop's exit); if a change of constants ever broke that, fail here rather than fold by an unwritten value.
            if (run.pipe_records) {
                if (run.cn > run.tail_len) return ZKHIP_ERR_ARG;
                ZK_TRY(run.tail_after_pipe());
                break;
            }
            continue;
        }
        int grid = 0;
        ZK_TRY(run.round_sums(&grid));
        run.close(run.d_partials, (uint32_t)grid);
    }
    return run.collect(h_lens, h_round_polys, h_challenges);
}

// Internal entries for gkr.hip (same shared object; not part of the C ABI): the two halves of prove_partial.  A caller chains the
// calls of ONE sumcheck on the device -- `cont` continues the transcript, `out_base` appends to the rounds already recorded --
// and reads all rounds back once (zk_multi_composed_collect), so nothing between the calls waits for the host.
// ex (nullable): the claimed sum from device memory instead of h_sum, an outer transcript to feed, and where the rounds and challenges
// are recorded instead of the context's small buffer (a caller that enqueues several sumchecks before it reads any of them back)
static_assert(CMP_OUTER_ROUNDS >= ZK_MAX_ROUNDS, "one ring slot per round");
int zk_multi_composed_enqueue(zkhip_ctx* c, const uint64_t* const* ptrs, const uint32_t* term_sizes, const uint64_t* const* lin_ptrs,
                              uint32_t n_terms, size_t n, const uint64_t* h_sum, int cont, uint32_t out_base, const ZkMcExtra* ex) {
    if (!c || !ptrs || !term_sizes || n_terms == 0 || n_terms > CMP_MAX_TERMS || (!h_sum && !cont && !(ex && ex->d_sum))) return ZKHIP_ERR_ARG;
    if (!is_pow2(n) || n < 2) return ZKHIP_ERR_SHAPE;
    const uint32_t n_vars = log2_exact(n);
    if (out_base + n_vars > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    ComposedRun run;
    ZK_TRY(run.setup(c, ptrs, term_sizes, n_terms, n, n_vars, 1, h_sum, 1, lin_ptrs, cont));
    run.out_base = out_base;
    if (ex) {
        run.sum_dev = ex->d_sum;
        run.outer.dev = (OuterDev*)ex->outer;
        run.outer.token = ex->token;
        if (ex->d_round_polys) run.d_rp = ex->d_round_polys;
        if (ex->d_challenges) run.d_ch = ex->d_challenges;
    }
    while (run.round < n_vars) {
        if (run.after() <= run.tail_len) {
            ZK_TRY(run.tail(run.current_tables(), (uint32_t)run.after(), run.folds()));
            break;
        }
        if (run.stage_ok()) { ZK_TRY(run.stage()); continue; }
        if (run.pipe_mid_ok()) {                 // one launch per round, one round ahead, down to the LDS tail
            ZK_TRY(run.pipe_mid());
            // pipe_mid() ends with the tables still unfolded at the last closed round's challenge and that challenge's Montgomery form
            // not yet in d_ch: only tail_after_pipe() continues from there.  It runs down to the LDS tail by construction (tail_len >=
            // 256 >= the steady loop's exit); if a change of constants ever broke that, fail here rather than fold by an unwritten value.
            if (run.pipe_records) {
                if (run.cn > run.tail_len) return ZKHIP_ERR_ARG;
                ZK_TRY(run.tail_after_pipe());
                break;
            }
            continue;
        }
        int grid = 0;
        ZK_TRY(run.round_sums(&grid));
        run.close(run.d_partials, (uint32_t)grid);
    }
    return run.check_launches();
}
int zk_multi_composed_collect(zkhip_ctx* c, uint32_t n_rounds, uint32_t* h_lens, uint64_t* h_round_polys, uint64_t* h_challenges) {
    return ComposedRun::collect_rounds(c, 1, 0, n_rounds, h_lens, h_round_polys, h_challenges);
}
// device addresses of the recorded challenges (4 u64 per round), for kernels that consume them without a host round trip
const uint64_t* zk_composed_challenges_dev(zkhip_ctx* c) { return c->small_u64(ZK_SMALL_CHALLENGES); }

// ComposedMultilinearTrait::element_wise_product (op 0) / element_wise_add (op 1), composed_multilinear.rs:105-119
extern "C" int zkhip_composed_element_wise(zkhip_ctx* c, int op, const uint64_t* const* ptrs, uint32_t k, size_t n, uint64_t* d_out) {
    if (!c || !ptrs || !d_out || op < 0 || op > 1) return ZKHIP_ERR_ARG;
    if (k == 0) return ZKHIP_ERR_INDEX;                    // self.polys[0] on an empty vector
    for (uint32_t q = 0; q < k; ++q) if (!ptrs[q]) return ZKHIP_ERR_ARG;
    if (n == 0) return ZKHIP_OK;
    ZK_TRY(c->activate());
    const int grid = mle_grid_stream(n);
    for (uint32_t q0 = 0; q0 < k; q0 += 8) {
        ElementwisePtrs t = {};
        const uint32_t cnt = std::min<uint32_t>(8, k - q0);
        for (uint32_t q = 0; q < cnt; ++q) t.in[q] = ptrs[q0 + q];
        if (op == 0)
            hipLaunchKernelGGL(composed_elementwise_kernel<true>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, t, cnt, n, q0 ? 1u : 0u, d_out);
        else
            hipLaunchKernelGGL(composed_elementwise_kernel<false>, dim3(grid), dim3(MLE_BLOCK), 0, c->stream, t, cnt, n, q0 ? 1u : 0u, d_out);
    }
    ZK_HIP(c, hipGetLastError());
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
// split-phase session: the same prover with its tables sharded over several GPUs (include/zkhip.h, zkhip_mc_*)
// ---------------------------------------------------------------------------------------
// The exchange shape of a session from its term sizes alone (what ComposedRun::setup derives): a rank that failed before it had a
// session still walks the protocol's exchange schedule with it (shard_protocol.hpp, "a failing rank must not hang its peers").
int zk_mc_shape(const uint32_t* sizes, uint32_t n_terms, uint32_t n_lin, uint32_t* rec, uint32_t* n_tables, uint32_t* tail_len, uint32_t* stage_vals) {
    if (!sizes || n_terms == 0 || n_terms > CMP_MAX_TERMS) return ZKHIP_ERR_ARG;
    uint32_t r = 0, total = 0;
    bool pairs = true;
    for (uint32_t p = 0; p < n_terms; ++p) {
        if (sizes[p] < 1 || sizes[p] > CMP_MAX_K) return ZKHIP_ERR_ARG;
        r += sizes[p] + 1;
        total += sizes[p];
        pairs = pairs && sizes[p] == 2;
    }
    if (r > CMP_MAX_REC) return ZKHIP_ERR_ARG;
    if (rec) *rec = r;
    if (n_tables) *n_tables = total + n_lin;
    if (tail_len) *tail_len = composed_tail_len(total + n_lin);
    if (stage_vals) *stage_vals = pairs ? n_terms * (uint32_t)CST_VALS : 0u;
    return ZKHIP_OK;
}

struct zkhip_mc_state {
    ComposedRun run;
    uint32_t world = 1;
    int sums_pending = 0;        // which record is out and not absorbed yet: 0 none, 1 a round record (zkhip_mc_round_sums), 2 a stage record
    size_t pending_cn = 0;       // entries per table when that record was taken (the matching absorb must find the session where it left it)
};

// the shape of a live session (what zk_mc_shape gives for its arguments) and its current entries per table
int zk_mc_state_shape(zkhip_mc_state* s, uint32_t* rec, uint32_t* n_tables, uint32_t* tail_len, uint32_t* stage_vals, size_t* n_local) {
    if (!s) return ZKHIP_ERR_ARG;
    const ComposedRun& run = s->run;
    ZK_TRY(zk_mc_shape(run.term_sizes, run.n_terms, run.total_all - run.total, rec, n_tables, tail_len, stage_vals));
    if (n_local) *n_local = run.after();
    return ZKHIP_OK;
}

extern "C" int zkhip_mc_begin_ex(zkhip_ctx* c, const uint64_t* const* d_local_tables, const uint32_t* term_sizes, uint32_t n_terms,
                                 const uint64_t* const* d_local_lin, size_t n_local, uint32_t world, int multi, const uint64_t* h_sum,
                                 int cont, uint32_t out_base, zkhip_mc_state** out) {
    if (!c || !d_local_tables || !term_sizes || !out) return ZKHIP_ERR_ARG;
    if (n_terms == 0 || n_terms > CMP_MAX_TERMS || (!multi && n_terms != 1) || (multi && !h_sum && !cont)) return ZKHIP_ERR_ARG;
    if ((cont || d_local_lin) && !multi) return ZKHIP_ERR_ARG;
    if (!is_pow2(n_local) || world == 0 || !is_pow2(world)) return ZKHIP_ERR_SHAPE;
    const uint32_t rounds = log2_exact(n_local) + log2_exact(world);
    if (rounds == 0 || out_base + rounds > ZK_MAX_ROUNDS) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    if (c->ws_lent) return ZKHIP_ERR_BUSY;      // one session per context: its tables live in the context's workspace
    zkhip_mc_state* s = new (std::nothrow) zkhip_mc_state();
    if (!s) return ZKHIP_ERR_NOMEM;
    s->world = world;
    const int rc = s->run.setup(c, d_local_tables, term_sizes, n_terms, n_local, rounds, multi, h_sum, 1, d_local_lin, cont);
    s->run.out_base = out_base;
    if (rc != ZKHIP_OK) {
        delete s;
        return rc;
    }
    if (world > s->run.tail_len) {   // shards of one entry per table must fit the replicated tail when gathered
        delete s;
        return ZKHIP_ERR_SHAPE;
    }
    c->ws_lent = true;              // until finish / abort: every other user of the workspace gets ZKHIP_ERR_BUSY
    *out = s;
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_begin(zkhip_ctx* c, const uint64_t* const* d_local_tables, const uint32_t* term_sizes, uint32_t n_terms,
                              size_t n_local, uint32_t world, int multi, const uint64_t* h_sum, zkhip_mc_state** out) {
    return zkhip_mc_begin_ex(c, d_local_tables, term_sizes, n_terms, nullptr, n_local, world, multi, h_sum, 0, 0, out);
}
extern "C" int zkhip_mc_record_len(zkhip_mc_state* s, uint32_t* rec, uint32_t* n_tables) {
    if (!s || !rec) return ZKHIP_ERR_ARG;
    *rec = s->run.meta.rec;
    if (n_tables) *n_tables = s->run.total_all;
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_local_len(zkhip_mc_state* s, size_t* n_now) {
    if (!s || !n_now) return ZKHIP_ERR_ARG;
    *n_now = s->run.after();
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_tail_capacity(zkhip_mc_state* s, uint32_t* cap) {
    if (!s || !cap) return ZKHIP_ERR_ARG;
    *cap = s->run.tail_len;
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_round_sums(zkhip_mc_state* s, uint64_t* d_out) {
    if (!s || !d_out) return ZKHIP_ERR_ARG;
    ComposedRun& run = s->run;
    if (s->sums_pending || run.round >= run.n_rounds || run.after() < 2) return ZKHIP_ERR_ARG;
    ZK_TRY(run.c->activate());
    int grid = 0;
    ZK_TRY(run.round_sums(&grid));
    hipLaunchKernelGGL(composed_reduce_kernel, dim3(1), dim3(MLE_BLOCK), 0, run.c->stream, run.d_partials, (uint32_t)grid, run.meta.rec, d_out);
    ZK_HIP(run.c, hipGetLastError());
    s->sums_pending = 1;
    s->pending_cn = run.cn;
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_absorb(zkhip_mc_state* s, const uint64_t* d_gathered, uint32_t world) {
    // a ROUND record only: a stage record (n_terms * 20 values) read with a round record's length would give a wrong transcript silently
    if (!s || !d_gathered || world != s->world || s->sums_pending != 1 || s->pending_cn != s->run.cn) return ZKHIP_ERR_ARG;
    ZK_TRY(s->run.c->activate());
    s->run.close(d_gathered, world);
    ZK_HIP(s->run.c, hipGetLastError());
    s->sums_pending = 0;
    return ZKHIP_OK;
}
// Two rounds per exchange (composed_stage.hpp) when every term is a product of two tables: *vals = the length of the stage record
// (n_terms * 20 field elements: the cross-block sums C[4][4] and the additive table's block sums L[4] of every term), or 0 when the
// session's next step cannot be a stage (a term with K != 2, a fold pending from a plain round, fewer than 4 local entries or fewer
// than 2 rounds left).
extern "C" int zkhip_mc_stage_record_len(zkhip_mc_state* s, uint32_t* vals) {
    if (!s || !vals) return ZKHIP_ERR_ARG;
    *vals = (!s->sums_pending && s->run.stage_possible(4)) ? s->run.n_terms * (uint32_t)CST_VALS : 0u;
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_stage_sums(zkhip_mc_state* s, uint64_t* d_out) {
    if (!s || !d_out) return ZKHIP_ERR_ARG;
    ComposedRun& run = s->run;
    if (s->sums_pending || !run.stage_possible(4)) return ZKHIP_ERR_ARG;
    ZK_TRY(run.c->activate());
    int grid = 0;
    ZK_TRY(run.stage_sums(&grid));
    hipLaunchKernelGGL(composed_stage_reduce_kernel, dim3(1), dim3(CST_BLOCK), 0, run.c->stream, run.d_partials, (uint32_t)grid, run.n_terms * (uint32_t)CST_VALS, d_out);
    ZK_HIP(run.c, hipGetLastError());
    s->sums_pending = 2;
    s->pending_cn = run.cn;
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_stage_absorb(zkhip_mc_state* s, const uint64_t* d_gathered, uint32_t world) {
    if (!s || !d_gathered || world != s->world || s->sums_pending != 2 || s->pending_cn != s->run.cn || !s->run.stage_possible(4)) return ZKHIP_ERR_ARG;
    ZK_TRY(s->run.c->activate());
    ZK_TRY(s->run.stage_close(d_gathered, world));
    ZK_HIP(s->run.c, hipGetLastError());
    s->sums_pending = 0;
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_local_tables(zkhip_mc_state* s, uint64_t* d_out) {
    if (!s || !d_out || s->sums_pending) return ZKHIP_ERR_ARG;
    ComposedRun& run = s->run;
    ZK_TRY(run.c->activate());
    const size_t m = run.after();
    const TailTables tt = run.current_tables();
    for (uint32_t q = 0; q < run.total_all; ++q) {
        uint64_t* dst = d_out + (size_t)q * m * 4;
        if (run.folds()) {   // the fold at the last challenge is still pending (the next round's kernel would have done it)
            hipLaunchKernelGGL(fold_kernel<false>, dim3(mle_grid_stream((m + 1) / 2)), dim3(MLE_BLOCK), 0, run.c->stream, tt.in[q], dst, m, log2_exact(m),
                               run.prev_challenge(), FrArg{}, (uint64_t*)nullptr);
        } else {
            ZK_HIP(run.c, hipMemcpyAsync(dst, tt.in[q], m * 32, hipMemcpyDeviceToDevice, run.c->stream));
        }
    }
    ZK_HIP(run.c, hipGetLastError());
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_tail(zkhip_mc_state* s, const uint64_t* d_tables, uint32_t m) {
    if (!s || !d_tables || s->sums_pending) return ZKHIP_ERR_ARG;
    ComposedRun& run = s->run;
    if (!is_pow2(m) || m > run.tail_len || m != run.after() * s->world || run.round + log2_exact(m) != run.n_rounds) return ZKHIP_ERR_SHAPE;
    ZK_TRY(run.c->activate());
    TailTables tt = {};
    for (uint32_t q = 0; q < run.total_all; ++q) tt.in[q] = d_tables + (size_t)q * m * 4;
    ZK_TRY(run.tail(tt, m, false));
    ZK_HIP(run.c, hipGetLastError());
    return ZKHIP_OK;
}
extern "C" int zkhip_mc_finish(zkhip_mc_state* s, uint32_t* h_lens, uint64_t* h_round_polys, uint64_t* h_challenges) {
    if (!s) return ZKHIP_ERR_ARG;
    int rc = ZKHIP_OK;
    if (h_round_polys || h_challenges) {
        if (!h_round_polys || !h_challenges || (s->run.multi && !h_lens) || s->run.round != s->run.n_rounds) rc = ZKHIP_ERR_ARG;
        else if ((rc = s->run.c->activate()) == ZKHIP_OK) rc = s->run.collect(h_lens, h_round_polys, h_challenges);
    }
    s->run.c->ws_lent = false;
    delete s;
    return rc;
}
extern "C" int zkhip_mc_abort(zkhip_mc_state* s) {
    if (!s) return ZKHIP_ERR_ARG;
    zkhip_ctx* c = s->run.c;
    int rc = ZKHIP_OK;
    if (c->activate() != ZKHIP_OK || hipStreamSynchronize(c->stream) != hipSuccess) rc = ZKHIP_ERR_HIP;
    c->ws_lent = false;
    delete s;
    return rc;
}

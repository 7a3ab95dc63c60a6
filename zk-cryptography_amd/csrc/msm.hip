// msm.hip -- C-ABI entry points of the KZG commit path (MSM over G1) and SRS generation.
// gfx950 only.  No CPU fallback: every entry point launches HIP kernels or fails.
#include "../../include/zkhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "host_util.hpp"
#include "host_g1.hpp"
#include "mle_kernels.hpp"
#include "msm_kernels.hpp"
#include "srs_kernels.hpp"

using namespace zk;

// ---------------------------------------------------------------------------------------
// KZG commit (MSM)
// ---------------------------------------------------------------------------------------
// Geometry of one pass (msm_kernels.hpp: MsmWin / MsmSet / MsmPlan): how every problem's 256 scalar bits are cut into digit windows,
// and the bucket sets behind them.
//   one problem:   uniform windows of c bits, c = 16 from 2^13 points on (one lane per bucket: the accumulate pass wants many short
//                  lists, so the window is as wide as the sort allows as soon as the bucket reduction is not the larger cost; measured
//                  with tools/perf_msm.py, ZKHIP_MSM_C sweep: below 2^13 every c ends at the ~1 ms latency floor of the reduction
//                  passes).  16 x 16 = 256: the top window has 15 significant bits, half full, never sparse.
//   shifted table: at 2^20 points thirteen digit windows of 20 / 19 bits on ONE bucket set of 2^19 buckets; widths by SRS size (msm_table_widths).
//   several:       every problem gets a width of its OWN, log2(n_j) - delta bits (lists of 2^delta .. 2^(delta+1) points: the accumulate pass
//                  takes as long as its longest lists, and every further bit doubles the buckets the reduction passes walk), as w = ceil(256 / c)
//                  windows of the two widths ceil(256 / w) and one less that add up to exactly 256 -- so no window of any problem is
//                  sparse (a 3-bit top window put an eighth of all points into each of 4 buckets: the heavy-bucket passes, 0.6 ms) and
//                  all problems share ONE pass of every kernel (MultilinearKZG::open at 2^20: twenty problems, ~0.45 M buckets).
// the geometries of the last few shapes (an opening builds its twenty-problem geometry twice per call otherwise: ~0.1 ms of host time each)
static int msm_geometry(const MsmProblems& pr, bool shared, size_t table_stride, std::shared_ptr<const MsmGeometry>* out) {
    struct Entry { MsmProblems pr; bool shared; size_t stride; std::shared_ptr<const MsmGeometry> geo; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry& e : cache)
        if (e.shared == shared && e.stride == table_stride && e.pr.n == pr.n && std::memcmp(e.pr.off, pr.off, sizeof(uint32_t) * (pr.n + 1)) == 0) {
            *out = e.geo;
            return ZKHIP_OK;
        }
    auto g = std::make_shared<MsmGeometry>();
    ZK_TRY(msm_build_geometry(pr, shared, table_stride, *g));
    if (cache.size() >= 8) cache.erase(cache.begin());
    cache.push_back({pr, shared, table_stride, g});
    *out = g;
    return ZKHIP_OK;
}
// d_table (nullable): shifted-SRS table in the internal layout, entry w * table_stride + i = 2^(20 w) * point i; then
// d_points_xy is not read and there is exactly one problem.
// A commit in two halves, so that a caller with several commits in a row (MultilinearKZG::open) can run the host epilogue
// of one while the GPU works on the next: msm_enqueue launches everything and the copy of the (set, term) points into
// pinned slot `slot` (workspace from byte offset ws_off; *ws_used = what it occupies), msm_finish waits for that copy
// and runs the epilogue.
struct MsmPending {
    std::shared_ptr<const MsmGeometry> geo;
    MsmProblems pr;
    size_t n_out = 0;
    int slot = 0;
};
static int msm_enqueue(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf, const uint64_t* d_scalars, size_t n,
                       const MsmProblems& pr_in, const uint32_t* d_table, size_t table_stride, size_t ws_off, int slot, MsmPending* pend,
                       size_t* ws_used) {
    std::shared_ptr<const MsmGeometry> geo_p;
    ZK_TRY(msm_geometry(pr_in, d_table != nullptr, table_stride, &geo_p));
    const MsmGeometry& geo = *geo_p;
    MsmProblems pr = pr_in;
    std::memcpy(pr.win_first, geo.win_first, sizeof(pr.win_first));
    MsmPlan pl = geo.pl;
    const size_t n_buckets = pl.n_buckets;
    const size_t n_segments = n_buckets / MSM_SEG;
    const size_t n_out = pl.n_terms;
    // workspace carve-up (all 256-byte aligned)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_counts = 0;
    const size_t o_offsets = o_counts + al(n_buckets * 4);
    const size_t o_order = o_offsets + al(n_buckets * 4);
    const size_t o_bins = o_order + al(n_buckets * 4);                  // the order pass's bins and, right behind them, the heavy lists' counters:
    const size_t o_ovf = o_bins + al(MSM_COUNT_BINS * 4);               //   ONE memset clears both
    const size_t o_sorted = o_ovf + al(sizeof(MsmOverflow));
    const size_t n_wgs = (n + SORT_TILE - 1) / SORT_TILE;
    const size_t o_items = o_sorted + al(geo.items * 4);
    const size_t o_wgc = o_items + al(geo.items * 8);
    const size_t o_pcnt = o_wgc + al(n_wgs * pl.n_parts * 4);
    const size_t o_poff = o_pcnt + al((pl.n_parts + 1) * 4);
    const size_t o_points = o_poff + al((pl.n_parts + 1) * 4);          // SRS in the internal 28-bit-limb layout
    const size_t o_buckets = o_points + al(d_table ? 0 : n * 128);
    const size_t o_segs = o_buckets + al(n_buckets * 256);
    const size_t o_sega = o_segs + al(n_segments * 256);
    const size_t o_terms = o_sega + al(n_segments * 256);
    // heavy buckets (msm_kernels.hpp pass 4b/4c): more than heavy_min points, so at most items / heavy_min of them;
    // a bucket of k points files ceil(k / MSM_HEAVY_REC) level-0 records and, above one record, a <= 256-way tree over them
    const uint32_t heavy_min = geo.heavy_min;
    const size_t items_max = geo.items;
    const size_t rec_cap = items_max / heavy_min + items_max / MSM_HEAVY_REC + 2;
    const size_t slots_cap = 3 * (items_max / MSM_HEAVY_REC) + 8;
    const size_t o_rc = o_terms + al(n_out * 192);                       // rows / columns of the segment sums (msm_rowcol_kernel)
    const size_t o_rec = o_rc + al((size_t)pl.n_rc * 256);
    const size_t o_part = o_rec + al(MSM_HEAVY_LEVELS * rec_cap * sizeof(MsmHeavyRec));
    const size_t total = o_part + al(slots_cap * 256);
    if (ws_used) *ws_used = total;
    if (!pend) return ZKHIP_OK;                     // size query only
    ZK_TRY(c->reserve_ws(ws_off + total));
    char* ws = (char*)c->d_ws + ws_off;
    uint32_t* counts = (uint32_t*)(ws + o_counts);
    uint32_t* offsets = (uint32_t*)(ws + o_offsets);
    uint32_t* sorted = (uint32_t*)(ws + o_sorted);
    uint32_t* order = (uint32_t*)(ws + o_order);
    uint32_t* bins = (uint32_t*)(ws + o_bins);
    const uint32_t* points_u = d_table ? d_table : (const uint32_t*)(ws + o_points);
    uint32_t* rc = (uint32_t*)(ws + o_rc);
    uint2* items = (uint2*)(ws + o_items);
    uint32_t* wg_counts = (uint32_t*)(ws + o_wgc);
    uint32_t* part_count = (uint32_t*)(ws + o_pcnt);
    uint32_t* part_off = (uint32_t*)(ws + o_poff);
    uint32_t* buckets = (uint32_t*)(ws + o_buckets);
    uint32_t* segs = (uint32_t*)(ws + o_segs);
    uint32_t* sega = (uint32_t*)(ws + o_sega);
    uint64_t* terms = (uint64_t*)(ws + o_terms);
    MsmOverflow* ovf = (MsmOverflow*)(ws + o_ovf);
    MsmHeavyRec* rec = (MsmHeavyRec*)(ws + o_rec);
    uint32_t* partials = (uint32_t*)(ws + o_part);

    {   // the geometry tables: device copy of this slot, uploaded when they differ from the last commit's
        const size_t b_wins = al(geo.wins.size() * sizeof(MsmWin)), b_sets = al(geo.sets.size() * sizeof(MsmSet));
        const size_t b_part = al(geo.part_set.size() * 2), b_rcwg = al(geo.rcwg_set.size() * 2), b_term = al(geo.termwg_set.size() * 2);
        const size_t bytes = b_wins + b_sets + b_part + b_rcwg + b_term;
        ZK_TRY(c->reserve_msm_tab(slot, bytes));
        char* dev = (char*)c->msm_tab_dev[slot];
        if (c->msm_tab_geo[slot].get() != (const void*)geo_p.get()) {
            char* pin = (char*)c->msm_tab_pin[slot];
            // the staging buffer may still feed the upload of the slot's previous commit only if that commit has not been waited for;
            // every caller ends (msm_finish / stream synchronize) a slot's commit before it enqueues the next one there
            std::memcpy(pin, geo.wins.data(), geo.wins.size() * sizeof(MsmWin));
            std::memcpy(pin + b_wins, geo.sets.data(), geo.sets.size() * sizeof(MsmSet));
            std::memcpy(pin + b_wins + b_sets, geo.part_set.data(), geo.part_set.size() * 2);
            std::memcpy(pin + b_wins + b_sets + b_part, geo.rcwg_set.data(), geo.rcwg_set.size() * 2);
            std::memcpy(pin + b_wins + b_sets + b_part + b_rcwg, geo.termwg_set.data(), geo.termwg_set.size() * 2);
            ZK_HIP(c, hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, c->stream));
            c->msm_tab_geo[slot] = geo_p;
        }
        pl.wins = (const MsmWin*)dev;
        pl.sets = (const MsmSet*)(dev + b_wins);
        pl.part_set = (const uint16_t*)(dev + b_wins + b_sets);
        pl.rcwg_set = (const uint16_t*)(dev + b_wins + b_sets + b_part);
        pl.termwg_set = (const uint16_t*)(dev + b_wins + b_sets + b_part + b_rcwg);
    }
    const int grid_n = (int)std::min<size_t>((n + MSM_BLOCK - 1) / MSM_BLOCK, 256 * 8);
    if (!d_table) {
        ProfScope ps(c, "msm_convert_points", 224.0 * (double)n);
        hipLaunchKernelGGL(msm_convert_points_kernel, dim3(grid_n), dim3(MSM_BLOCK), 0, c->stream, d_points_xy, n, (uint32_t*)(ws + o_points));
    }
    {   // two-level counting sort of the (point, window) pairs by bucket; also yields counts[] and offsets[]
        ProfScope ps(c, "msm_sort", 32.0 * (double)n);
        hipLaunchKernelGGL(msm_sort_count_kernel, dim3((unsigned)n_wgs), dim3(MSM_BLOCK), 0, c->stream, d_scalars, d_points_inf, n, pl, pr, wg_counts);
        hipLaunchKernelGGL(msm_sort_bases_kernel, dim3((pl.n_parts + MSM_BLOCK / 64 - 1) / (MSM_BLOCK / 64)), dim3(MSM_BLOCK), 0, c->stream, wg_counts,
                           (uint32_t)n_wgs, pl.n_parts, part_count);
        hipLaunchKernelGGL(msm_sort_part_scan_kernel, dim3(1), dim3(1024), 0, c->stream, part_count, pl.n_parts, part_off);
        hipLaunchKernelGGL(msm_sort_scatter_kernel, dim3((unsigned)n_wgs), dim3(MSM_BLOCK), 0, c->stream, d_scalars, d_points_inf, n, pl, pr,
                           wg_counts, part_off, items);
        // level 2 also takes the histogram of the bucket counts (the order pass) and files the heavy buckets: both cleared here
        ZK_HIP(c, hipMemsetAsync(bins, 0, (size_t)(o_sorted - o_bins), c->stream));      // bins + MsmOverflow
        hipLaunchKernelGGL(msm_sort_local_kernel, dim3(pl.n_parts), dim3(SORT_LOCAL_BLOCK), 0, c->stream, items, part_off, pl, sorted, counts, offsets,
                           heavy_min, ovf, rec, (uint32_t)rec_cap, bins);
    }
    {
        // heavy buckets (none with uniform scalars: four empty launches): filed by the sort and summed by passes 4b / 4c IN
        // FRONT of the accumulate pass (on a second stream beside it they did not find a free slot for most of its duration -- its
        // first workgroups walk the longest lists and hold every register of the chip)
        ProfScope ps(c, "msm_overflow", 0.0);
        hipLaunchKernelGGL(msm_heavy_points_kernel, dim3(512) /* 64 KiB of LDS each: two per CU are resident, the records are walked in a loop */, dim3(MSM_BLOCK), MSM_BLOCK * 256, c->stream, points_u, sorted, ovf, rec,
                           partials, buckets);
        for (int level = 1; level < MSM_HEAVY_LEVELS; ++level)
            hipLaunchKernelGGL(msm_heavy_tree_kernel, dim3(level == 1 ? 256 : 16), dim3(MSM_BLOCK), MSM_BLOCK * 256, c->stream, ovf,
                               (uint32_t)level, rec + (size_t)level * rec_cap, partials, buckets);
    }
    {   // bucket order by descending point count
        ProfScope ps(c, "msm_order", 0.0);
        const unsigned gb = (unsigned)((n_buckets + MSM_BLOCK - 1) / MSM_BLOCK);
        hipLaunchKernelGGL(msm_order_scan_kernel, dim3(1), dim3(1024), 0, c->stream, bins);
        hipLaunchKernelGGL(msm_order_scatter_kernel, dim3(gb), dim3(MSM_BLOCK), 0, c->stream, counts, (uint32_t)n_buckets, bins, order);
    }
    {
        ProfScope ps(c, "msm_accumulate", 128.0 * (double)n);
        hipLaunchKernelGGL(msm_accumulate_kernel, dim3((unsigned)((n_buckets + MSM_BLOCK - 1) / MSM_BLOCK)), dim3(MSM_BLOCK), 0,
                           c->stream, points_u, sorted, offsets, counts, order, (uint32_t)n_buckets, heavy_min, buckets);
    }
    {
        ProfScope ps(c, "msm_segment", 0.0);
        hipLaunchKernelGGL(msm_segment_kernel, dim3((unsigned)((n_segments + MSM_BLOCK - 1) / MSM_BLOCK)), dim3(MSM_BLOCK), 0,
                           c->stream, buckets, (uint32_t)n_segments, segs, sega);
    }
    {
        ProfScope ps(c, "msm_terms", 0.0);
        // one wave per workgroup: lines and trees are summed inside a wave, without LDS
        hipLaunchKernelGGL(msm_rowcol_kernel, dim3(pl.n_rcwg), dim3(64), 0, c->stream, segs, sega, pl, rc);
        hipLaunchKernelGGL(msm_rowcol_terms_kernel, dim3(pl.n_termwg), dim3(64), 0, c->stream, rc, pl, terms);
    }
    ZK_HIP(c, hipGetLastError());
    ZK_TRY(c->reserve_msm_pin(slot, n_out * 192));
    ZK_HIP(c, hipMemcpyAsync(c->msm_pin[slot], terms, n_out * 192, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipEventRecord(c->msm_ev[slot], c->stream));
    pend->geo = geo_p;
    pend->pr = pr;
    pend->n_out = n_out;
    pend->slot = slot;
    return ZKHIP_OK;
}
static int msm_finish(zkhip_ctx* c, const MsmPending& pend, uint64_t* h_out_xy, uint8_t* h_out_inf) {
    ZK_HIP(c, hipEventSynchronize(c->msm_ev[pend.slot]));
    const MsmGeometry& geo = *pend.geo;
    const MsmProblems& pr = pend.pr;
    const uint64_t* h_terms = (const uint64_t*)c->msm_pin[pend.slot];
    // host epilogue, per problem: sum over its (set, term) points of 2^exp * point
    auto finish = [&](uint32_t j) {
        std::vector<zkhost::Xyzz> pts;
        std::vector<uint32_t> exps;
        for (uint32_t s = geo.prob_set_first[j]; s < geo.prob_set_first[j + 1]; ++s) {
            const uint32_t n_terms = (geo.sets[s].bits & 0xffu) - MSM_SEG_LOG;      // 1 + n_bits
            for (uint32_t t = 0; t < n_terms; ++t) {
                zkhost::Xyzz p;
                std::memcpy(&p, &h_terms[24 * ((size_t)geo.sets[s].term_base + t)], 192);
                pts.push_back(p);
                exps.push_back(geo.set_exp[s] + (t == 0 ? 0 : MSM_SEG_LOG + (t - 1)));
            }
        }
        zkhost::Xyzz res = zkhost::weighted_sum_pow2(pts, exps);
        h_out_inf[j] = zkhost::xyzz_to_affine(res, h_out_xy + 12 * (size_t)j) ? 0 : 1;
    };
    // the chains of different problems are independent (~0.25 ms each): a batch spreads them over the context's host threads, the
    // largest problems (most points) first
    ZkHostPool* pool = pr.n > 1 ? c->pool() : nullptr;
    if (!pool) {
        for (uint32_t j = 0; j < pr.n; ++j) finish(j);
    } else {
        pool->run(pr.n, [&](unsigned j) { finish(j); });
    }
    return ZKHIP_OK;
}
static int msm_commit_multi(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf, const uint64_t* d_scalars, size_t n,
                            const MsmProblems& pr, uint64_t* h_out_xy, uint8_t* h_out_inf, const uint32_t* d_table = nullptr,
                            size_t table_stride = 0) {
    MsmPending pend;
    ZK_TRY(msm_enqueue(c, d_points_xy, d_points_inf, d_scalars, n, pr, d_table, table_stride, 0, 0, &pend, nullptr));
    return msm_finish(c, pend, h_out_xy, h_out_inf);
}
static int msm_commit(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf, const uint64_t* d_scalars, size_t n,
                      uint64_t* h_out_xy, uint8_t* h_out_inf) {
    MsmProblems one = {};
    one.n = 1;
    one.off[1] = (uint32_t)n;
    return msm_commit_multi(c, d_points_xy, d_points_inf, d_scalars, n, one, h_out_xy, h_out_inf);
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
    return msm_commit(c, d_points_xy, d_points_inf, d_scalars, n, h_out_xy, h_out_inf);
}

// A table begins with a 128-byte HEADER (magic, kind, the number of points it was built for, its window widths, its number of entries):
// the layout of a table is a function of its size AND of a tuning variable read once per process (msm_level_table_widths), the kernels address
// it blindly, and a table built for another size -- or by a process that ran with another value -- would give a silently wrong commitment.
// zkhip_kzg_commit_table / _commit_begin / zkhip_kzg_open_tables compare the header with the geometry they are about to use (one 128-byte read
// the first time a (table, size) pair is seen in the process, remembered until a table is built at that address again) -> ZKHIP_ERR_ARG on a
// mismatch.  (Not re-read: a buffer the library did not build, placed at a remembered address.)
constexpr size_t ZK_TABLE_HEADER_BYTES = 128;
constexpr uint64_t ZK_TABLE_MAGIC = 0x31304c42544b5a00ull;       // "\0ZKTBL01"
enum : uint32_t { ZK_TABLE_SHIFTED_SRS = 1, ZK_TABLE_LEVELS = 2 };
struct ZkTableHeader {
    uint64_t magic;
    uint32_t kind, version;
    uint64_t n_points, entries;
    uint32_t W, hi, n_hi;          // the widths of the (first, largest) table
    uint32_t widths_hash;          // FNV-1a over (h, W, hi, n_hi) of every level (level tables)
    uint8_t pad_[ZK_TABLE_HEADER_BYTES - 48];
};
static_assert(sizeof(ZkTableHeader) == ZK_TABLE_HEADER_BYTES, "one 128-byte entry in front of the table");
// tables whose header has been compared with the geometry of (kind, n_points): one list per PROCESS (contexts are per thread and share
// SRS objects), forgotten whenever a table is built at that address again
namespace {
struct TableOk { const void* p; size_t n; uint32_t kind; int device; };
std::mutex g_table_ok_mutex;
std::vector<TableOk> g_table_ok;
bool table_remembered(const zkhip_ctx* c, const void* p, size_t n, uint32_t kind) {
    std::lock_guard<std::mutex> lk(g_table_ok_mutex);
    for (const auto& t : g_table_ok) if (t.p == p && t.n == n && t.kind == kind && t.device == c->device) return true;
    return false;
}
void table_remember(const zkhip_ctx* c, const void* p, size_t n, uint32_t kind) {
    std::lock_guard<std::mutex> lk(g_table_ok_mutex);
    if (g_table_ok.size() >= 64) g_table_ok.erase(g_table_ok.begin());
    g_table_ok.push_back(TableOk{p, n, kind, c->device});
}
void table_forget(const zkhip_ctx* c, const void* p) {
    std::lock_guard<std::mutex> lk(g_table_ok_mutex);
    for (size_t i = g_table_ok.size(); i-- > 0;) if (g_table_ok[i].p == p && g_table_ok[i].device == c->device) g_table_ok.erase(g_table_ok.begin() + i);
}
}  // namespace
// the owner is about to free (or reuse) the buffer of a shifted-SRS table or of level tables: its address is no longer a known table
extern "C" int zkhip_table_release(zkhip_ctx* c, const void* d_table) {
    if (!c || !d_table) return ZKHIP_ERR_ARG;
    table_forget(c, d_table);
    return ZKHIP_OK;
}
static size_t level_tables_first(size_t n_points, size_t* lvl_off);
static ZkTableHeader table_header_for(uint32_t kind, size_t n_points) {
    ZkTableHeader h;
    std::memset(&h, 0, sizeof(h));
    h.magic = ZK_TABLE_MAGIC; h.kind = kind; h.version = 1; h.n_points = n_points;
    uint32_t fnv = 2166136261u;
    auto mix = [&](uint32_t v) { for (int b = 0; b < 4; ++b) { fnv ^= (v >> (8 * b)) & 0xffu; fnv *= 16777619u; } };
    if (kind == ZK_TABLE_SHIFTED_SRS) {
        const MsmLevelWidths lw = msm_table_widths(n_points);
        h.W = lw.W; h.hi = lw.hi; h.n_hi = lw.n_hi; h.entries = (uint64_t)lw.W * n_points;
        mix((uint32_t)n_points); mix(lw.W); mix(lw.hi); mix(lw.n_hi);
    } else {
        const size_t first = level_tables_first(n_points, nullptr);
        for (size_t q = first; q >= 1; q /= 2) {
            const MsmLevelWidths lw = msm_level_table_widths(q, 2 * first - 1);
            if (q == first) { h.W = lw.W; h.hi = lw.hi; h.n_hi = lw.n_hi; }
            h.entries += (uint64_t)lw.W * q;
            mix((uint32_t)q); mix(lw.W); mix(lw.hi); mix(lw.n_hi);
        }
    }
    h.widths_hash = fnv;
    return h;
}
static int table_write_header(zkhip_ctx* c, void* d_table, uint32_t kind, size_t n_points) {
    const ZkTableHeader h = table_header_for(kind, n_points);
    ZK_HIP(c, hipMemcpy(d_table, &h, sizeof(h), hipMemcpyHostToDevice));      // (once per SRS; synchronous)
    table_remember(c, d_table, n_points, kind);
    return ZKHIP_OK;
}
// -> the table proper (behind the header), or nullptr with *rc set
static const uint32_t* table_check(zkhip_ctx* c, const void* d_table, uint32_t kind, size_t n_points, int* rc) {
    *rc = ZKHIP_OK;
    if (!table_remembered(c, d_table, n_points, kind)) {
        ZkTableHeader got;
        if (hipMemcpy(&got, d_table, sizeof(got), hipMemcpyDeviceToHost) != hipSuccess) { *rc = ZKHIP_ERR_HIP; return nullptr; }
        const ZkTableHeader want = table_header_for(kind, n_points);
        if (std::memcmp(&got, &want, 48) != 0) { *rc = ZKHIP_ERR_ARG; return nullptr; }     // not a table, another kind, another size, other widths
        table_remember(c, d_table, n_points, kind);
    }
    return (const uint32_t*)((const char*)d_table + ZK_TABLE_HEADER_BYTES);
}

// ---------------------------------------------------------------------------------------
// commits in flight: a commit is a throughput-bound accumulate pass followed by latency-bound reduction passes (chains of
// ~25 us point additions on a mostly idle chip) and a host epilogue; a prover that commits several polynomials in a row
// hides the latter behind the next commits' accumulate passes.  Three slots (2 / 3 / 4 in flight: 3.03 / 2.91 / 2.98 ms per 2^20-point
// commit; the accumulate pass and the sort are throughput work, only the reductions and the epilogue hide); same results as the synchronous calls.
// ---------------------------------------------------------------------------------------
extern "C" int zkhip_kzg_commit_begin(zkhip_ctx* c, const uint64_t* d_points_xy, const void* d_table_with_header, const uint8_t* d_points_inf,
                                      size_t n_points, const uint64_t* d_scalars, size_t n_scalars, int require_equal_len,
                                      uint32_t* ticket) {
    if (!c || !ticket || (!d_points_xy == !d_table_with_header)) return ZKHIP_ERR_ARG;
    if (require_equal_len && n_points != n_scalars) return ZKHIP_ERR_SHAPE;   // multilinear_kzg.rs:36-41
    if (n_scalars > n_points) return ZKHIP_ERR_INDEX;                          // univariate_kzg.rs:53
    const size_t n = n_scalars;
    if (n == 0 || !d_scalars) return ZKHIP_ERR_ARG;                           // nothing to overlap: use the synchronous call
    if (n >= ((size_t)1 << 31) || (d_table_with_header && n_points * msm_table_widths(n_points).W >= ((size_t)1 << 31))) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    const uint32_t* d_table = nullptr;
    if (d_table_with_header) {
        int trc = ZKHIP_OK;
        d_table = table_check(c, d_table_with_header, ZK_TABLE_SHIFTED_SRS, n_points, &trc);       // built for THESE n_points, with the widths used below?
        if (!d_table) return trc;
    }
    int slot = -1;
    bool any = false;
    for (int k = zkhip_ctx::ASYNC_SLOTS - 1; k >= 0; --k) { if (!c->async_pend[k]) slot = k; else any = true; }
    if (slot < 0) return ZKHIP_ERR_BUSY;
    MsmProblems one = {};
    one.n = 1;
    one.off[1] = (uint32_t)n;
    size_t used = 0;
    ZK_TRY(msm_enqueue(c, nullptr, nullptr, nullptr, n, one, (const uint32_t*)d_table, n_points, 0, 0, nullptr, &used));
    used = (used + 4095) & ~(size_t)4095;
    if (!any) {
        if (c->ws_lent) return ZKHIP_ERR_BUSY;
        ZK_TRY(c->reserve_ws(zkhip_ctx::ASYNC_SLOTS * used));
        c->async_region = std::max(used, c->ws_bytes / zkhip_ctx::ASYNC_SLOTS & ~(size_t)4095);
    } else if (used > c->async_region) {
        return ZKHIP_ERR_BUSY;            // a larger commit than the one in flight: end that one first
    }
    ZK_TRY(c->ensure_side_streams());
    MsmPending* pend = new (std::nothrow) MsmPending();
    if (!pend) return ZKHIP_ERR_NOMEM;
    hipStream_t const main_stream = c->stream;
    int rc = ZKHIP_OK;
    if (hipEventRecord(c->fork_ev, main_stream) != hipSuccess || hipStreamWaitEvent(c->side[slot], c->fork_ev, 0) != hipSuccess) rc = ZKHIP_ERR_HIP;
    if (rc == ZKHIP_OK) {
        const bool lent = c->ws_lent;
        c->ws_lent = false;               // the reservation above covers both regions; msm_enqueue's own reserve is a no-op
        c->stream = c->side[slot];
        rc = msm_enqueue(c, d_points_xy, d_points_inf, d_scalars, n, one, (const uint32_t*)d_table, n_points, (size_t)slot * c->async_region, slot,
                         pend, nullptr);
        c->stream = main_stream;
        c->ws_lent = lent;
    }
    if (rc != ZKHIP_OK) {
        // a partially enqueued commit may have kernels running on the side stream: drain them before the slot, its workspace
        // region and its pinned buffer are handed out again
        hipStreamSynchronize(c->side[slot]);
        delete pend;
        return rc;
    }
    c->async_pend[slot] = pend;
    c->ws_lent = true;                    // until the last commit in flight has ended
    *ticket = (uint32_t)slot;
    return ZKHIP_OK;
}
extern "C" int zkhip_kzg_commit_end(zkhip_ctx* c, uint32_t ticket, uint64_t* h_out_xy, uint8_t* h_out_inf) {
    if (!c || ticket >= (uint32_t)zkhip_ctx::ASYNC_SLOTS || !c->async_pend[ticket]) return ZKHIP_ERR_ARG;
    MsmPending* pend = (MsmPending*)c->async_pend[ticket];
    int rc = c->activate();
    if (rc == ZKHIP_OK && h_out_xy && h_out_inf) rc = msm_finish(c, *pend, h_out_xy, h_out_inf);
    else if (hipStreamSynchronize(c->side[ticket]) != hipSuccess) rc = ZKHIP_ERR_HIP;            // abandoned: just drain it
    if (hipStreamWaitEvent(c->stream, c->msm_ev[ticket], 0) != hipSuccess && rc == ZKHIP_OK) rc = ZKHIP_ERR_HIP;   // workspace reuse stays ordered
    delete pend;
    c->async_pend[ticket] = nullptr;
    bool any = false;
    for (int k = 0; k < zkhip_ctx::ASYNC_SLOTS; ++k) any = any || c->async_pend[k];
    if (!any) c->ws_lent = false;
    return rc;
}

// ---------------------------------------------------------------------------------------
// SRS fingerprint: the first two and last two points with their infinity flags, in ONE launch and ONE copy
// ---------------------------------------------------------------------------------------
static __global__ void srs_fingerprint_kernel(const uint64_t* __restrict__ xy, const uint8_t* __restrict__ inf, size_t n, uint64_t* __restrict__ out) {
    const uint32_t t = threadIdx.x;                 // 52 threads: 4 points x (12 words + 1 flag)
    if (t >= 52) return;
    const uint32_t k = t / 13, w = t - 13 * k;
    const size_t idx[4] = {(size_t)0, n > 1 ? (size_t)1 : (size_t)0, n > 1 ? n - 2 : (size_t)0, n - 1};
    out[t] = w < 12 ? xy[12 * idx[k] + w] : (inf ? (uint64_t)inf[idx[k]] : 0);
}
extern "C" int zkhip_srs_fingerprint(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf, size_t n_points, uint64_t* h_out) {
    if (!c || !d_points_xy || !h_out) return ZKHIP_ERR_ARG;
    if (n_points == 0) { std::memset(h_out, 0, 52 * 8); return ZKHIP_OK; }
    ZK_TRY(c->activate());
    if (!c->d_fingerprint) ZK_HIP(c, hipMalloc(&c->d_fingerprint, 512));      // a buffer of its own: commits in flight hold the workspace
    if (!c->guard_stream) {
        // a stream of the highest priority: with commits in flight the one-wave kernel takes the next slot that frees instead of queueing
        // behind the thousands of workgroups of an accumulate pass (109 us per call on the caller's stream, rocprofv3 of the bench's in-flight leg)
        int least = 0, greatest = 0;
        ZK_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        ZK_HIP(c, hipStreamCreateWithPriority(&c->guard_stream, hipStreamNonBlocking, greatest));
        ZK_HIP(c, hipEventCreateWithFlags(&c->guard_ev, hipEventDisableTiming));
    }
    uint64_t* pin = c->pinned_u64(ZK_PIN_PTS);      // (evaluation points' staging area: no prover runs during this call)
    static_assert(ZK_PIN_PROOF - ZK_PIN_PTS >= 52, "staging area too small");
    ZK_HIP(c, hipEventRecord(c->guard_ev, c->stream));                         // behind whatever the caller's stream still does to the SRS
    ZK_HIP(c, hipStreamWaitEvent(c->guard_stream, c->guard_ev, 0));
    hipLaunchKernelGGL(srs_fingerprint_kernel, dim3(1), dim3(64), 0, c->guard_stream, d_points_xy, d_points_inf, n_points, (uint64_t*)c->d_fingerprint);
    ZK_HIP(c, hipMemcpyAsync(pin, c->d_fingerprint, 52 * 8, hipMemcpyDeviceToHost, c->guard_stream));
    ZK_HIP(c, hipStreamSynchronize(c->guard_stream));
    std::memcpy(h_out, pin, 52 * 8);
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// shifted-SRS table: commitments with one bucket set for all windows
// ---------------------------------------------------------------------------------------
extern "C" size_t zkhip_srs_table_bytes(size_t n_points) { return n_points ? ZK_TABLE_HEADER_BYTES + n_points * msm_table_widths(n_points).W * 128 : 0; }

// windows 0 .. n_windows-1 of c bits over n affine points: table entry w * n + i = 2^(c w) * point i (internal 28-bit-limb layout, 128 bytes);
// the context's workspace holds one window in XYZZ and affine form meanwhile
static int build_shift_table(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf, size_t n, uint32_t width, uint32_t n_windows,
                             uint32_t* table, uint32_t n_wide = ~0u /* windows >= n_wide are width - 1 bits */) {
    // workspace: XYZZ of one window (192 n) | affine (96 n) | infinity flags (n)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_xyzz = 0, o_aff = al(192 * n), o_inf = o_aff + al(96 * n);
    ZK_TRY(c->reserve_ws(o_inf + al(n)));
    char* ws = (char*)c->d_ws;
    const unsigned grid = (unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK);
    hipLaunchKernelGGL(msm_convert_points_kernel, dim3(std::min<unsigned>(grid, 256 * 8)), dim3(MSM_BLOCK), 0, c->stream, d_points_xy, n, table);
    if (d_points_inf) {   // entries of points at infinity: all-zero coordinates (their scalars are skipped by the sort anyway)
        hipLaunchKernelGGL(msm_clear_inf_kernel, dim3(grid), dim3(MSM_BLOCK), 0, c->stream, d_points_inf, n, table);
    }
    const size_t n_threads = (n + SRS_CHUNK - 1) / SRS_CHUNK;
    for (uint32_t w = 1; w < n_windows; ++w) {
        const uint32_t* prev = table + (size_t)(w - 1) * n * 32;
        uint32_t* cur = table + (size_t)w * n * 32;
        hipLaunchKernelGGL(msm_shift_points_kernel, dim3(grid), dim3(MSM_BLOCK), 0, c->stream, prev, n, w - 1 < n_wide ? width : width - 1,
                           (uint64_t*)(ws + o_xyzz));
        hipLaunchKernelGGL(srs_batch_affine_kernel, dim3((unsigned)((n_threads + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0, c->stream,
                           (const uint64_t*)(ws + o_xyzz), n, (uint64_t*)(ws + o_aff), (uint8_t*)(ws + o_inf));
        hipLaunchKernelGGL(msm_convert_points_kernel, dim3(std::min<unsigned>(grid, 256 * 8)), dim3(MSM_BLOCK), 0, c->stream,
                           (const uint64_t*)(ws + o_aff), n, cur);
    }
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}
extern "C" int zkhip_srs_precompute(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf, size_t n,
                                    void* d_table) {
    if (!c || !d_points_xy || !d_table) return ZKHIP_ERR_ARG;
    if (n == 0) return ZKHIP_OK;
    const MsmLevelWidths lw = msm_table_widths(n);
    if (n * lw.W >= ((size_t)1 << 31)) return ZKHIP_ERR_SHAPE;   // entry index + sign bit in 32 bits
    ZK_TRY(c->activate());
    table_forget(c, d_table);
    ZK_TRY(build_shift_table(c, d_points_xy, d_points_inf, n, lw.hi, lw.W, (uint32_t*)((char*)d_table + ZK_TABLE_HEADER_BYTES), lw.n_hi));
    ZK_HIP(c, hipStreamSynchronize(c->stream));   // the workspace is reused by the next call
    return table_write_header(c, d_table, ZK_TABLE_SHIFTED_SRS, n);      // last: a table whose build failed carries no valid header
}

// LEVEL TABLES: shifted tables of the folded SRS levels MultilinearKZG::open commits against in ONE batch (the levels of at most
// OPEN_BATCH_MAX points), each with the window widths of msm_level_table_widths, end to end in level order (msm_geometry.hpp).
constexpr size_t OPEN_BATCH_MAX_DEFAULT = (size_t)1 << 19;
constexpr size_t MSM_SMALL_MAX = 4096;          // commits / openings of at most this many scalars take the short path (msm_small_batch)
static size_t level_tables_first(size_t n_points, size_t* lvl_off) {   // size of the first (largest) level that has a table; *lvl_off = its offset in the folded array
    size_t h = n_points / 2, off = 0;
    while (h > OPEN_BATCH_MAX_DEFAULT) { off += h; h /= 2; }
    if (lvl_off) *lvl_off = off;
    return h;
}
extern "C" size_t zkhip_srs_level_tables_bytes(size_t n_points) {
    if (n_points < 2 || !is_pow2(n_points)) return 0;
    size_t entries = 0;
    const size_t first = level_tables_first(n_points, nullptr);
    for (size_t h = first; h >= 1; h /= 2) entries += (size_t)msm_level_table_widths(h, 2 * first - 1).W * h;
    return ZK_TABLE_HEADER_BYTES + entries * 128;
}
extern "C" int zkhip_srs_level_tables(zkhip_ctx* c, const uint64_t* d_folded_xy, const uint8_t* d_folded_inf, size_t n_points, void* d_tables_with_header) {
    if (!c || !d_folded_xy || !d_folded_inf || !d_tables_with_header) return ZKHIP_ERR_ARG;
    if (n_points < 2 || !is_pow2(n_points)) return ZKHIP_ERR_SHAPE;
    if (zkhip_srs_level_tables_bytes(n_points) / 128 >= ((size_t)1 << 31)) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    table_forget(c, d_tables_with_header);
    void* d_tables = (char*)d_tables_with_header + ZK_TABLE_HEADER_BYTES;
    size_t off = 0, entry = 0;
    const size_t first = level_tables_first(n_points, &off);
    if (n_points <= MSM_SMALL_MAX) {
        // a small SRS: every level in one go (msm_small_level_windows_kernel) -- a few ms instead of 0.3-0.4 s, so that a small opening can
        // build its tables on first use
        MsmSmallTabArgs a = {};
        for (size_t h = first; h >= 1; h /= 2) {
            const MsmLevelWidths lw = msm_level_table_widths(h, 2 * first - 1);
            const uint32_t j = a.n_levels++;
            a.h[j] = (uint32_t)h; a.pt_off[j] = a.total_points; a.tab_off[j] = (uint32_t)entry; a.W[j] = lw.W; a.hi[j] = lw.hi; a.n_hi[j] = lw.n_hi;
            a.total_points += (uint32_t)h;
            entry += (size_t)lw.W * h;
        }
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t o_pts = 0, o_xyzz = al(128 * (size_t)a.total_points), o_aff = o_xyzz + al(192 * entry), o_inf = o_aff + al(96 * entry);
        ZK_TRY(c->reserve_ws(o_inf + al(entry)));
        char* ws = (char*)c->d_ws;
        uint32_t* d_pts = (uint32_t*)(ws + o_pts);
        hipLaunchKernelGGL(msm_convert_points_kernel, dim3((a.total_points + MSM_BLOCK - 1) / MSM_BLOCK), dim3(MSM_BLOCK), 0, c->stream, d_folded_xy + 12 * off,
                           (size_t)a.total_points, d_pts);
        hipLaunchKernelGGL(msm_clear_inf_kernel, dim3((a.total_points + MSM_BLOCK - 1) / MSM_BLOCK), dim3(MSM_BLOCK), 0, c->stream, d_folded_inf + off,
                           (size_t)a.total_points, d_pts);
        hipLaunchKernelGGL(msm_small_level_windows_kernel, dim3((a.total_points + 63) / 64), dim3(64), 0, c->stream, a, (const uint32_t*)d_pts, (uint64_t*)(ws + o_xyzz));
        const size_t n_threads = (entry + SRS_CHUNK - 1) / SRS_CHUNK;
        hipLaunchKernelGGL(srs_batch_affine_kernel, dim3((unsigned)((n_threads + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0, c->stream,
                           (const uint64_t*)(ws + o_xyzz), entry, (uint64_t*)(ws + o_aff), (uint8_t*)(ws + o_inf));
        const unsigned grid = (unsigned)((entry + MSM_BLOCK - 1) / MSM_BLOCK);
        hipLaunchKernelGGL(msm_convert_points_kernel, dim3(std::min<unsigned>(grid, 256 * 8)), dim3(MSM_BLOCK), 0, c->stream, (const uint64_t*)(ws + o_aff), entry,
                           (uint32_t*)d_tables);
        hipLaunchKernelGGL(msm_clear_inf_kernel, dim3(grid), dim3(MSM_BLOCK), 0, c->stream, (const uint8_t*)(ws + o_inf), entry, (uint32_t*)d_tables);
        ZK_HIP(c, hipGetLastError());
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        return table_write_header(c, d_tables_with_header, ZK_TABLE_LEVELS, n_points);
    }
    for (size_t h = first; h >= 1; h /= 2) {
        const MsmLevelWidths lw = msm_level_table_widths(h, 2 * first - 1);      // the batch: the levels first, first / 2, ..., 1
        const uint32_t W = lw.W;
        ZK_TRY(build_shift_table(c, d_folded_xy + 12 * off, d_folded_inf + off, h, lw.hi, W, (uint32_t*)d_tables + entry * 32, lw.n_hi));
        off += h;
        entry += (size_t)W * h;
    }
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    return table_write_header(c, d_tables_with_header, ZK_TABLE_LEVELS, n_points);
}

// Commits of at most MSM_SMALL_MAX scalars against a shifted-SRS table: the plane sums of msm_kernels.hpp "commits of a few thousand
// points" -- two launches, one copy, a host epilogue of <= 20 doublings.  ZKHIP_MSM_SMALL=0 keeps the bucket pipeline (A/B runs).
// problem j: n[j] scalars from sc_off[j] on against the table at tab_off[j] (entries) of stride[j] points with the widths lw[j]; results
// h_out_xy[12 j], h_out_inf[j]
struct MsmSmallProblem { size_t n, stride, tab_off, sc_off; MsmLevelWidths lw; };
static int msm_small_batch(zkhip_ctx* c, const uint32_t* d_table, const uint8_t* d_inf, const uint64_t* d_scalars, const MsmSmallProblem* pr,
                           uint32_t nprob, uint64_t* h_out_xy, uint8_t* h_out_inf) {
    if (nprob == 0 || nprob > (uint32_t)MSM_SMALL_PROBS) return ZKHIP_ERR_SHAPE;
    MsmSmallArgs a = {};
    a.table = d_table; a.scalars = d_scalars; a.inf = d_inf; a.nprob = nprob;
    size_t total_pairs = 0;
    for (uint32_t j = 0; j < nprob; ++j) {
        a.n[j] = (uint32_t)pr[j].n; a.stride[j] = (uint32_t)pr[j].stride; a.W[j] = pr[j].lw.W; a.hi[j] = pr[j].lw.hi; a.n_hi[j] = pr[j].lw.n_hi;
        a.tab_off[j] = (uint32_t)pr[j].tab_off; a.sc_off[j] = (uint32_t)pr[j].sc_off;
        a.planes = std::max(a.planes, pr[j].lw.hi);
        total_pairs += pr[j].n * pr[j].lw.W;
    }
    // ~2 waves per SIMD in all, dealt to the problems by their share of the pairs (at least one slot each, at most a wave per 64 pairs)
    const size_t budget = std::max<size_t>(1, 2048 / a.planes);
    for (uint32_t j = 0; j < nprob; ++j) {
        const size_t pairs = pr[j].n * pr[j].lw.W, chunks = (pairs + 63) / 64;
        const size_t share = total_pairs ? (budget * pairs + total_pairs - 1) / total_pairs : 1;
        a.slot_first[j + 1] = a.slot_first[j] + (uint32_t)std::max<size_t>(1, std::min(chunks, share));
    }
    a.total_slots = a.slot_first[nprob];
    const size_t part_bytes = (size_t)a.planes * a.total_slots * 256, terms_bytes = (size_t)nprob * a.planes * 192;
    ZK_TRY(c->reserve_ws(part_bytes + 256 + terms_bytes));
    a.partials = (uint32_t*)c->d_ws;
    a.terms = (uint64_t*)((char*)c->d_ws + ((part_bytes + 255) & ~(size_t)255));
    {
        ProfScope ps(c, "msm_small", 128.0 * (double)total_pairs / std::max<uint32_t>(1, a.W[0]));
        hipLaunchKernelGGL(msm_small_planes_kernel, dim3(a.total_slots, a.planes), dim3(64), 0, c->stream, a);
        hipLaunchKernelGGL(msm_small_reduce_kernel, dim3(a.planes, nprob), dim3(64), 0, c->stream, a);
    }
    ZK_HIP(c, hipGetLastError());
    ZK_TRY(c->reserve_msm_pin(0, terms_bytes));
    ZK_HIP(c, hipMemcpyAsync(c->msm_pin[0], a.terms, terms_bytes, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    const char* pin = (const char*)c->msm_pin[0];
    auto finish = [&](uint32_t j) {                                  // the weighted sum of a problem's plane sums: <= 20 doublings
        std::vector<zkhost::Xyzz> pts(a.hi[j]);
        std::vector<uint32_t> exps(a.hi[j]);
        for (uint32_t t = 0; t < a.hi[j]; ++t) {
            std::memcpy(&pts[t], pin + 192 * ((size_t)j * a.planes + t), 192);
            exps[t] = t;
        }
        const zkhost::Xyzz res = zkhost::weighted_sum_pow2(pts, exps);
        h_out_inf[j] = zkhost::xyzz_to_affine(res, h_out_xy + 12 * (size_t)j) ? 0 : 1;
    };
    ZkHostPool* pool = nprob > 1 ? c->pool() : nullptr;
    if (!pool) {
        for (uint32_t j = 0; j < nprob; ++j) finish(j);
    } else {
        pool->run(nprob, [&](unsigned j) { finish(j); });
    }
    return ZKHIP_OK;
}
static int msm_commit_small(zkhip_ctx* c, const uint32_t* d_table, size_t stride, const uint8_t* d_inf, const uint64_t* d_scalars, size_t n,
                            uint64_t* h_out_xy, uint8_t* h_out_inf) {
    const MsmSmallProblem one = {n, stride, 0, 0, msm_table_widths(stride)};
    return msm_small_batch(c, d_table, d_inf, d_scalars, &one, 1, h_out_xy, h_out_inf);
}
static bool msm_small_on() {
    static const bool on = [] { const char* e = std::getenv("ZKHIP_MSM_SMALL"); return !e || std::atoi(e) != 0; }();
    return on;
}

extern "C" int zkhip_kzg_commit_table(zkhip_ctx* c, const void* d_table_with_header, const uint8_t* d_points_inf, size_t n_points,
                                      const uint64_t* d_scalars, size_t n_scalars, int require_equal_len, uint64_t* h_out_xy,
                                      uint8_t* h_out_inf) {
    if (!c || !h_out_xy || !h_out_inf) return ZKHIP_ERR_ARG;
    if (require_equal_len && n_points != n_scalars) return ZKHIP_ERR_SHAPE;   // multilinear_kzg.rs:36-41
    if (n_scalars > n_points) return ZKHIP_ERR_INDEX;                          // univariate_kzg.rs:53
    const size_t n = n_scalars;
    if (n == 0) { std::memset(h_out_xy, 0, 96); *h_out_inf = 1; return ZKHIP_OK; }
    if (!d_table_with_header || !d_scalars) return ZKHIP_ERR_ARG;
    if (n_points * msm_table_widths(n_points).W >= ((size_t)1 << 31)) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    int trc = ZKHIP_OK;
    const uint32_t* d_table = table_check(c, d_table_with_header, ZK_TABLE_SHIFTED_SRS, n_points, &trc);     // built for THESE n_points, with the widths used below?
    if (!d_table) return trc;
    if (msm_small_on() && n <= MSM_SMALL_MAX) return msm_commit_small(c, (const uint32_t*)d_table, n_points, d_points_inf, d_scalars, n, h_out_xy, h_out_inf);
    MsmProblems one = {};
    one.n = 1;
    one.off[1] = (uint32_t)n;
    return msm_commit_multi(c, nullptr, d_points_inf, d_scalars, n, one, h_out_xy, h_out_inf, (const uint32_t*)d_table, n_points);
}

extern "C" int zkhip_msm_geometry_info(const size_t* h_offsets, uint32_t n_problems, uint16_t* h_win_first, uint8_t* h_win_bits,
                                       uint32_t* h_totals) {
    if (!h_offsets || !h_win_first || !h_win_bits || !h_totals) return ZKHIP_ERR_ARG;
    if (n_problems == 0 || n_problems > (uint32_t)MSM_MAX_PROBLEMS) return ZKHIP_ERR_SHAPE;
    MsmProblems pr = {};
    pr.n = n_problems;
    for (uint32_t j = 0; j <= n_problems; ++j) {
        if (h_offsets[j] >= ((size_t)1 << 31) || (j && h_offsets[j] < h_offsets[j - 1])) return ZKHIP_ERR_SHAPE;
        pr.off[j] = (uint32_t)(h_offsets[j] - h_offsets[0]);
    }
    MsmGeometry g;
    ZK_TRY(msm_build_geometry(pr, false, 0, g));
    for (uint32_t j = 0; j <= n_problems; ++j) h_win_first[j] = g.win_first[j];
    for (size_t v = 0; v < g.wins.size(); ++v) h_win_bits[v] = (uint8_t)(g.wins[v].bits & 0xffu);
    const uint32_t totals[8] = {g.pl.n_wins, g.pl.n_sets, g.pl.n_buckets, g.pl.n_parts, g.pl.n_terms, g.pl.n_rcwg, g.pl.n_termwg, g.heavy_min};
    std::memcpy(h_totals, totals, sizeof(totals));
    return ZKHIP_OK;
}

extern "C" int zkhip_kzg_commit_batch(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf,
                                      const uint64_t* d_scalars, const size_t* h_offsets, uint32_t n_problems, uint64_t* h_out_xy,
                                      uint8_t* h_out_inf) {
    if (!c || !h_offsets || !h_out_xy || !h_out_inf) return ZKHIP_ERR_ARG;
    if (n_problems == 0) return ZKHIP_OK;
    if (n_problems > (uint32_t)MSM_MAX_PROBLEMS) return ZKHIP_ERR_SHAPE;
    MsmProblems pr = {};
    pr.n = n_problems;
    for (uint32_t j = 0; j <= n_problems; ++j) {
        if (h_offsets[j] >= ((size_t)1 << 31) || (j && h_offsets[j] < h_offsets[j - 1])) return ZKHIP_ERR_SHAPE;
        pr.off[j] = (uint32_t)(h_offsets[j] - h_offsets[0]);
    }
    const size_t n = pr.off[n_problems];
    if (n == 0) {
        for (uint32_t j = 0; j < n_problems; ++j) { std::memset(h_out_xy + 12 * (size_t)j, 0, 96); h_out_inf[j] = 1; }
        return ZKHIP_OK;
    }
    if (!d_points_xy || !d_scalars) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    const size_t o = h_offsets[0];
    return msm_commit_multi(c, d_points_xy + 12 * o, d_points_inf ? d_points_inf + o : nullptr, d_scalars + 4 * o, n, pr, h_out_xy, h_out_inf);
}

// ---------------------------------------------------------------------------------------
// MultilinearKZG::open
// ---------------------------------------------------------------------------------------
// folded SRS levels S_0 .. S_{nv-1} (n - 1 affine points); d_tmp_xyzz: (n - 1) x 192 bytes of scratch
static int fold_srs_levels(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf, size_t n, uint64_t* d_tmp_xyzz,
                           uint64_t* d_out_xy, uint8_t* d_out_inf) {
    size_t h = n / 2, off = 0;
    hipLaunchKernelGGL(srs_fold_affine_kernel, dim3((unsigned)((h + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0, c->stream,
                       d_points_xy, d_points_inf, h, d_tmp_xyzz);
    while (h > 1) {
        const size_t next = off + h;
        h /= 2;
        hipLaunchKernelGGL(srs_fold_xyzz_kernel, dim3((unsigned)((h + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0, c->stream,
                           d_tmp_xyzz + 24 * off, h, d_tmp_xyzz + 24 * next);
        off = next;
    }
    const size_t total = n - 1, n_threads = (total + SRS_CHUNK - 1) / SRS_CHUNK;
    hipLaunchKernelGGL(srs_batch_affine_kernel, dim3((unsigned)((n_threads + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0, c->stream,
                       d_tmp_xyzz, total, d_out_xy, d_out_inf);
    ZK_HIP(c, hipGetLastError());
    return ZKHIP_OK;
}

extern "C" int zkhip_srs_fold_levels(zkhip_ctx* c, const uint64_t* d_points_xy, const uint8_t* d_points_inf, size_t n_points,
                                     uint64_t* d_out_xy, uint8_t* d_out_inf) {
    if (!c || !d_points_xy || !d_out_xy || !d_out_inf) return ZKHIP_ERR_ARG;
    if (n_points < 2 || !is_pow2(n_points)) return ZKHIP_ERR_SHAPE;
    ZK_TRY(c->activate());
    ZK_TRY(c->reserve_ws((n_points - 1) * 192));
    return fold_srs_levels(c, d_points_xy, d_points_inf, n_points, (uint64_t*)c->d_ws, d_out_xy, d_out_inf);
}

extern "C" int zkhip_kzg_open(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_points, size_t n_eval_points,
                              const uint64_t* d_points_xy, const uint8_t* d_points_inf, size_t n_points,
                              const uint64_t* d_folded_xy, const uint8_t* d_folded_inf, uint64_t* h_evaluation,
                              uint64_t* h_proofs_xy, uint8_t* h_proofs_inf) {
    return zkhip_kzg_open_tables(c, d_evals, n, h_points, n_eval_points, d_points_xy, d_points_inf, n_points, d_folded_xy, d_folded_inf, nullptr,
                                 h_evaluation, h_proofs_xy, h_proofs_inf);
}
extern "C" int zkhip_kzg_open_tables(zkhip_ctx* c, const uint64_t* d_evals, size_t n, const uint64_t* h_points, size_t n_eval_points,
                                     const uint64_t* d_points_xy, const uint8_t* d_points_inf, size_t n_points,
                                     const uint64_t* d_folded_xy, const uint8_t* d_folded_inf, const void* d_level_tables_with_header,
                                     uint64_t* h_evaluation, uint64_t* h_proofs_xy, uint8_t* h_proofs_inf) {
    if (!c || !d_evals || !h_points || !d_points_xy || !h_evaluation || !h_proofs_xy || !h_proofs_inf) return ZKHIP_ERR_ARG;
    if (d_level_tables_with_header && !d_folded_inf) return ZKHIP_ERR_ARG;      // the tables belong to cached folded levels
    if (!is_pow2(n)) return ZKHIP_ERR_SHAPE;
    const uint32_t n_vars = log2_exact(n);
    if (n_eval_points != n_vars) return ZKHIP_ERR_SHAPE;   // evaluation_form.rs:163-167
    if (n_points != n) return ZKHIP_ERR_SHAPE;             // multilinear_kzg.rs:36-41
    if (n_vars < 2) return ZKHIP_ERR_SHAPE;                // :73 `variable_index - 1` underflows
    if (n >= ((size_t)1 << 31)) return ZKHIP_ERR_SHAPE;
    if ((d_folded_xy == nullptr) != (d_folded_inf == nullptr)) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    const void* d_level_tables = nullptr;
    if (d_level_tables_with_header) {
        int trc = ZKHIP_OK;
        d_level_tables = table_check(c, d_level_tables_with_header, ZK_TABLE_LEVELS, n_points, &trc);   // the level tables of an SRS of THIS size, with the widths used below?
        if (!d_level_tables) return trc;
    }
    // aux layout: quotients of all rounds (n - 1, laid out like the folded SRS) | remainder ping (n/2) | pong (n/4) |
    // [folded SRS xy, inf]
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_q = 0, o_ping = o_q + al(n * 32), o_pong = o_ping + al(n / 2 * 32);
    const size_t o_fxy = o_pong + al((n / 4 + 1) * 32), o_finf = o_fxy + al((n - 1) * 96);
    const bool own_fold = d_folded_xy == nullptr;
    ZK_TRY(c->reserve_aux(own_fold ? o_finf + al(n - 1) : o_fxy));
    char* aux = (char*)c->d_aux;
    if (own_fold) {
        ZK_TRY(c->reserve_ws((n - 1) * 192));
        ProfScope ps(c, "open_fold_srs", 0.0);
        ZK_TRY(fold_srs_levels(c, d_points_xy, d_points_inf, n, (uint64_t*)c->d_ws, (uint64_t*)(aux + o_fxy), (uint8_t*)(aux + o_finf)));
        d_folded_xy = (const uint64_t*)(aux + o_fxy);
        d_folded_inf = (const uint8_t*)(aux + o_finf);
    }
    uint64_t* d_q = (uint64_t*)(aux + o_q);
    // All quotients first (a chain of n_vars small kernels on the caller's stream; none of them depends on a commit).  Then the commits:
    // every round with at most 2^19 quotient entries -- all twenty of a 2^20 opening -- as problems of ONE batched commit, each with a
    // window width of its own (msm_build_geometry), on the caller's stream: one pass of every MSM kernel, each filling the chip, one
    // bucket reduction.  Larger rounds (openings above 2^20) are commits of their own, two in flight on side streams beside the batch.
    // (History.  Rounds 1-3: every round above 2^14 as a commit pipeline of its own, five beside each other on three hardware queues, each
    // paying the full reduction of 2^19 buckets and every kernel running at 2-3 x its stand-alone duration beside the others' accumulate
    // passes: 8.5-9.2 ms at 2^20 against 3.2 ms for ONE commit of as many points (profiles/r03/open_trace_pipelines.txt).  Then three
    // batches of one width each (14 / 12 / 10 bits) beside each other, fronts lined up: 6.7 ms -- their sparse top windows needed the
    // heavy-bucket passes (0.6 ms) and the three reductions ended 0.9 ms after the last accumulate pass.  ZKHIP_OPEN_PIPELINES=1 runs the
    // rounds above 2^14 as pipelines again for an A/B.)
    const bool pipelines = [] { const char* e = std::getenv("ZKHIP_OPEN_PIPELINES"); return e && e[0] == '1'; }();
    size_t OPEN_BATCH_MAX = pipelines ? (size_t)1 << 14 : OPEN_BATCH_MAX_DEFAULT;
    if (const char* e = std::getenv("ZKHIP_OPEN_BATCH_LOG")) {   // tuning aid (tools/perf_open.py)
        const int v = std::atoi(e);
        if (v >= 8 && v <= 20) OPEN_BATCH_MAX = (size_t)1 << v;
    }
    if (OPEN_BATCH_MAX != OPEN_BATCH_MAX_DEFAULT) d_level_tables = nullptr;    // the tables are laid out for the default batch
    // result slot / stream of the single commits (they rotate over the slots); the batch has the last slot
    int NSLOT = pipelines ? 5 : 2;
    if (const char* e = std::getenv("ZKHIP_OPEN_SLOTS")) {   // tuning aid (tools/perf_open.py): fewer single commits beside each other
        const int v = std::atoi(e);
        if (v >= 1 && v <= NSLOT) NSLOT = v;
    }
    const int sl_batch = zkhip_ctx::MSM_SLOTS - 1;
    const uint64_t* cur = d_evals;
    size_t cn = n, lvl_off = 0;
    MsmProblems batch = {};
    size_t batch_first_off = 0;
    uint32_t batch_first_round = 0;
    struct Large { uint32_t round; size_t off, h; };
    std::vector<Large> large;
    // (a small opening against its tables -- the short path below -- takes all its rounds in ONE launch: a dozen 5 us launches otherwise)
    const bool small_open = d_level_tables && n <= MSM_SMALL_MAX && OPEN_BATCH_MAX == OPEN_BATCH_MAX_DEFAULT && msm_small_on() && n_vars <= (uint32_t)ZK_MAX_ROUNDS;
    if (small_open) {
        PtsArg zp = {};
        std::memcpy(zp.v, h_points, 32 * (size_t)n_vars);
        hipLaunchKernelGGL(open_steps_small_kernel, dim3(1), dim3(MLE_BLOCK), 0, c->stream, d_evals, (uint32_t)n, zp, n_vars, d_q, (uint64_t*)(aux + o_ping),
                           (uint64_t*)(aux + o_pong));
    }
    for (uint32_t i = 0; i < n_vars; ++i) {
        FrArg z = {};
        std::memcpy(z.v, h_points + 4 * (size_t)i, 32);
        uint64_t* rem = (uint64_t*)(aux + ((i & 1) ? o_pong : o_ping));
        if (!small_open)
            hipLaunchKernelGGL(open_step_kernel, dim3(mle_grid_stream(cn / 2)), dim3(MLE_BLOCK), 0, c->stream, cur, cn, z, d_q + 4 * lvl_off, rem);
        const size_t h = cn / 2;   // |q_i| = |S_i|
        if (h > OPEN_BATCH_MAX) {
            large.push_back({i, lvl_off, h});
        } else {                   // the batched rounds follow each other, so the problems lie end to end
            if (batch.n == 0) { batch_first_off = lvl_off; batch_first_round = i; }
            batch.off[batch.n] = (uint32_t)(lvl_off - batch_first_off);
            batch.off[++batch.n] = (uint32_t)(lvl_off + h - batch_first_off);
        }
        lvl_off += h;
        cur = rem;
        cn = h;
    }
    ZK_HIP(c, hipGetLastError());
    if (small_open && large.empty() && batch.n && batch.n <= (uint32_t)MSM_SMALL_PROBS) {
        // a small opening against its level tables: every round's quotient commit is a plain sum per digit bit (msm_small_batch) --
        // two launches for all rounds instead of the bucket pipeline's seventeen (0.96 -> 0.6x ms at 2^12)
        MsmSmallProblem sp[MSM_SMALL_PROBS];
        const size_t total = batch.off[batch.n];
        size_t entry = 0;
        for (uint32_t j = 0; j < batch.n; ++j) {
            const size_t h = batch.off[j + 1] - batch.off[j];
            const MsmLevelWidths lw = msm_level_table_widths(h, total);          // as zkhip_srs_level_tables laid the level out
            sp[j] = {h, h, entry, batch_first_off + batch.off[j], lw};
            entry += (size_t)lw.W * h;
        }
        ZK_TRY(msm_small_batch(c, (const uint32_t*)d_level_tables, d_folded_inf, d_q, sp, batch.n, h_proofs_xy + 12 * (size_t)batch_first_round,
                               h_proofs_inf + batch_first_round));
        ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), cur, 32, hipMemcpyDeviceToHost, c->stream));
        ZK_HIP(c, hipStreamSynchronize(c->stream));
        std::memcpy(h_evaluation, c->pinned_u64(ZK_PIN_RES), 32);
        return ZKHIP_OK;
    }
    // workspace: region k (k < NSLOT) is sized for the k-th single commit and reused by the commits k + NSLOT, k + 2 NSLOT, ... (each
    // half the size of its predecessor in the region or less); the batch has a region of its own.  Reserved once, up front: a
    // commit that grew the workspace later would move it under the commits in flight.
    size_t region_off[zkhip_ctx::MSM_SLOTS + 1] = {};
    for (int k = 0; k < NSLOT; ++k) {
        size_t used = 0;
        if ((size_t)k < large.size()) {
            MsmProblems one = {};
            one.n = 1;
            one.off[1] = (uint32_t)large[k].h;
            ZK_TRY(msm_enqueue(c, nullptr, nullptr, nullptr, large[k].h, one, nullptr, 0, 0, 0, nullptr, &used));
        }
        region_off[k + 1] = region_off[k] + ((used + 4095) & ~(size_t)4095);
    }
    {
        size_t used = 0;
        if (batch.n) ZK_TRY(msm_enqueue(c, nullptr, nullptr, nullptr, lvl_off - batch_first_off, batch, (const uint32_t*)d_level_tables, 0, 0, 0, nullptr, &used));
        ZK_TRY(c->reserve_ws(region_off[NSLOT] + used));
    }
    ZK_TRY(c->ensure_side_streams());
    ZK_HIP(c, hipEventRecord(c->fork_ev, c->stream));
    MsmPending pend[zkhip_ctx::MSM_SLOTS];
    int pend_round[zkhip_ctx::MSM_SLOTS];
    for (int k = 0; k < zkhip_ctx::MSM_SLOTS; ++k) pend_round[k] = -1;
    auto finish_slot = [&](int sl) -> int {
        if (pend_round[sl] < 0) return ZKHIP_OK;
        const int r = pend_round[sl];
        pend_round[sl] = -1;
        return msm_finish(c, pend[sl], h_proofs_xy + 12 * (size_t)r, h_proofs_inf + r);
    };
    hipStream_t const main_stream = c->stream;
    int rc = ZKHIP_OK;
    if (batch.n) {
        // on the caller's stream (nothing else runs there meanwhile); pipelines: on a side stream, and FIRST -- its chain of passes is the longest
        hipStream_t on = pipelines ? c->side[sl_batch] : main_stream;
        if (on != main_stream && hipStreamWaitEvent(on, c->fork_ev, 0) != hipSuccess) rc = ZKHIP_ERR_HIP;
        if (rc == ZKHIP_OK) {
            c->stream = on;                                // msm_enqueue launches on the context's stream
            rc = msm_enqueue(c, d_folded_xy + 12 * batch_first_off, d_folded_inf + batch_first_off, d_q + 4 * batch_first_off,
                             lvl_off - batch_first_off, batch, (const uint32_t*)d_level_tables, 0, region_off[NSLOT], sl_batch, &pend[sl_batch], nullptr);
            c->stream = main_stream;
            if (rc == ZKHIP_OK) pend_round[sl_batch] = (int)batch_first_round;
        }
    }
    for (size_t j = 0; j < large.size() && rc == ZKHIP_OK; ++j) {
        const int sl = (int)(j % NSLOT);
        if ((rc = finish_slot(sl)) != ZKHIP_OK) break;     // commit j - NSLOT used this region, stream and result slot
        MsmProblems one = {};
        one.n = 1;
        one.off[1] = (uint32_t)large[j].h;
        if (hipStreamWaitEvent(c->side[sl], c->fork_ev, 0) != hipSuccess) { rc = ZKHIP_ERR_HIP; break; }
        c->stream = c->side[sl];
        rc = msm_enqueue(c, d_folded_xy + 12 * large[j].off, d_folded_inf + large[j].off, d_q + 4 * large[j].off, large[j].h, one, nullptr, 0,
                         region_off[sl], sl, &pend[sl], nullptr);
        c->stream = main_stream;
        if (rc == ZKHIP_OK) pend_round[sl] = (int)large[j].round;
    }
    // the remaining epilogues, each as soon as its commit's results have landed; every slot is drained even after an error, so that
    // nothing is left running on a side stream
    for (;;) {
        int left = 0, ready = -1;
        for (int k = 0; k < zkhip_ctx::MSM_SLOTS; ++k) {
            if (pend_round[k] < 0) continue;
            ++left;
            if (ready < 0 && hipEventQuery(c->msm_ev[k]) != hipErrorNotReady) ready = k;
        }
        if (!left) break;
        if (ready < 0) { std::this_thread::yield(); continue; }
        const int r2 = finish_slot(ready);
        if (rc == ZKHIP_OK) rc = r2;
    }
    if (rc != ZKHIP_OK) {
        for (int k = 0; k < zkhip_ctx::MSM_SLOTS; ++k) hipStreamSynchronize(c->side[k]);
        return rc;
    }
    // the last remainder is poly(z): `evaluation`, and what the reference checks it against (:84-86)
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), cur, 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_evaluation, c->pinned_u64(ZK_PIN_RES), 32);
    return ZKHIP_OK;
}

// ---------------------------------------------------------------------------------------
// UnivariateKZG::open
// ---------------------------------------------------------------------------------------
extern "C" int zkhip_univariate_kzg_open(zkhip_ctx* c, const uint64_t* d_coeffs, size_t n, const uint64_t* h_z,
                                         const uint64_t* d_points_xy, const uint8_t* d_points_inf, size_t n_points,
                                         uint64_t* h_evaluation, uint64_t* h_proof_xy, uint8_t* h_proof_inf) {
    if (!c || !h_z || !h_evaluation || !h_proof_xy || !h_proof_inf || (n && !d_coeffs)) return ZKHIP_ERR_ARG;
    std::memset(h_proof_xy, 0, 96);
    if (n == 0) {                                   // the zero polynomial: evaluate() = 0; numerator [z] has degree 0 < 1 -> quotient zero
        std::memset(h_evaluation, 0, 32);
        *h_proof_inf = 1;
        return ZKHIP_OK;
    }
    if (n - 1 > n_points) return ZKHIP_ERR_INDEX;  // srs.powers_of_tau_in_g1[i] out of bounds (univariate_kzg.rs:75)
    if (n >= ((size_t)1 << 31)) return ZKHIP_ERR_SHAPE;
    if (n > 1 && !d_points_xy) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    const size_t per_block = (size_t)HS_T * HS_L;
    const uint32_t n_blocks = (uint32_t)((n + per_block - 1) / per_block);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_q = 0, o_vals = al(32 * n), o_carry = o_vals + al(32 * (size_t)n_blocks), o_eval = o_carry + al(32 * (size_t)n_blocks);
    ZK_TRY(c->reserve_aux(o_eval + 256));
    char* aux = (char*)c->d_aux;
    uint64_t* d_q = (uint64_t*)(aux + o_q);
    uint64_t* d_vals = (uint64_t*)(aux + o_vals);
    uint64_t* d_carry = (uint64_t*)(aux + o_carry);
    uint64_t* d_eval = (uint64_t*)(aux + o_eval);
    FrArg z = {};
    std::memcpy(z.v, h_z, 32);
    {
        ProfScope ps(c, "horner_scan", 96.0 * (double)n);
        hipLaunchKernelGGL(horner_scan_kernel<false>, dim3(n_blocks), dim3(HS_T), 0, c->stream, d_coeffs, n, z, d_vals, nullptr, nullptr, nullptr);
        hipLaunchKernelGGL(horner_top_kernel, dim3(1), dim3(1024), 0, c->stream, d_vals, n_blocks, z, d_carry, (uint64_t*)nullptr);
        hipLaunchKernelGGL(horner_scan_kernel<true>, dim3(n_blocks), dim3(HS_T), 0, c->stream, d_coeffs, n, z, nullptr, d_carry, d_q, d_eval);
    }
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_eval, 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_evaluation, c->pinned_u64(ZK_PIN_RES), 32);
    if (n == 1) { *h_proof_inf = 1; return ZKHIP_OK; }          // degree 0 < 1: quotient zero, proof = G1::default()
    return msm_commit(c, d_points_xy, d_points_inf, d_q, n - 1, h_proof_xy, h_proof_inf);
}

// DenseUnivariatePolynomial::evaluate (dense_univariate.rs:184-196) and ::degree (:199-207) on a device coefficient vector
extern "C" int zkhip_dense_evaluate(zkhip_ctx* c, const uint64_t* d_coeffs, size_t n, const uint64_t* h_z, uint64_t* h_out) {
    if (!c || !h_z || !h_out || (n && !d_coeffs)) return ZKHIP_ERR_ARG;
    if (n == 0) { std::memset(h_out, 0, 32); return ZKHIP_OK; }
    ZK_TRY(c->activate());
    const size_t per_block = (size_t)HS_T * HS_L;
    const uint32_t n_blocks = (uint32_t)((n + per_block - 1) / per_block);
    ZK_TRY(c->reserve_aux(64 * (size_t)n_blocks + 512));
    uint64_t* d_vals = (uint64_t*)c->d_aux;
    uint64_t* d_carry = d_vals + 4 * (size_t)n_blocks;
    uint64_t* d_eval = d_carry + 4 * (size_t)n_blocks;
    FrArg z = {};
    std::memcpy(z.v, h_z, 32);
    hipLaunchKernelGGL(horner_scan_kernel<false>, dim3(n_blocks), dim3(HS_T), 0, c->stream, d_coeffs, n, z, d_vals, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(horner_top_kernel, dim3(1), dim3(1024), 0, c->stream, d_vals, n_blocks, z, d_carry, d_eval);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_eval, 32, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_out, c->pinned_u64(ZK_PIN_RES), 32);
    return ZKHIP_OK;
}
extern "C" int zkhip_dense_degree(zkhip_ctx* c, const uint64_t* d_coeffs, size_t n, size_t* h_degree) {
    if (!c || !h_degree || (n && !d_coeffs)) return ZKHIP_ERR_ARG;
    *h_degree = 0;
    if (n == 0) return ZKHIP_OK;
    ZK_TRY(c->activate());
    unsigned long long* d_out = (unsigned long long*)c->small_u64(ZK_SMALL_RES);
    ZK_HIP(c, hipMemsetAsync(d_out, 0, 8, c->stream));
    hipLaunchKernelGGL(dense_degree_kernel, dim3(mle_grid(n)), dim3(MLE_BLOCK), 0, c->stream, d_coeffs, n, d_out);
    ZK_HIP(c, hipGetLastError());
    ZK_HIP(c, hipMemcpyAsync(c->pinned_u64(ZK_PIN_RES), d_out, 8, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(c, hipStreamSynchronize(c->stream));
    *h_degree = (size_t)c->pinned_u64(ZK_PIN_RES)[0];
    return ZKHIP_OK;
}

// the generator's window table of this context (built once: 32 chains of 255 additions, ~10 ms)
static int srs_gen_table(zkhip_ctx* c, const uint64_t** table_xy) {
    const size_t n_tab = (size_t)SRS_WINDOWS * SRS_DIGITS;
    if (!c->d_gen_table) {
        void* mem = nullptr;
        if (hipMalloc(&mem, n_tab * 97 + 256 + n_tab * 192) != hipSuccess) return ZKHIP_ERR_NOMEM;
        uint64_t* xy = (uint64_t*)mem;
        uint8_t* inf = (uint8_t*)mem + n_tab * 96;                 // never set: no multiple d * 2^(8w) * G with d < 256 is the identity
        uint64_t* xyzz = (uint64_t*)((char*)mem + ((n_tab * 97 + 255) & ~(size_t)255));
        hipLaunchKernelGGL(srs_gen_table_kernel, dim3(1), dim3(64), 0, c->stream, xyzz);
        const size_t n_threads = (n_tab + SRS_CHUNK - 1) / SRS_CHUNK;
        hipLaunchKernelGGL(srs_batch_affine_kernel, dim3((unsigned)((n_threads + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0,
                           c->stream, xyzz, n_tab, xy, inf);
        if (hipGetLastError() != hipSuccess) { (void)hipFree(mem); return ZKHIP_ERR_HIP; }
        c->d_gen_table = mem;
    }
    *table_xy = (const uint64_t*)c->d_gen_table;
    return ZKHIP_OK;
}

// scalars (device, n x 4) -> affine SRS points
static int srs_from_scalars(zkhip_ctx* c, const uint64_t* d_scalars, size_t n, uint64_t* d_out_xy, uint8_t* d_out_inf) {
    // workspace layout: [scalars n*32 (owned by caller region)] ... we only need n*192 for XYZZ here
    uint64_t* xyzz = (uint64_t*)((char*)c->d_ws + ((n * 32 + 255) & ~(size_t)255));
    const uint64_t* table = nullptr;
    ZK_TRY(srs_gen_table(c, &table));
    hipLaunchKernelGGL(srs_fixed_base_window_kernel, dim3((unsigned)((n + SRS_BLOCK - 1) / SRS_BLOCK)), dim3(SRS_BLOCK), 0, c->stream,
                       d_scalars, n, table, xyzz);
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
    PtsArg tau = {};
    if (n_vars) std::memcpy(tau.v, h_tau, 32 * (size_t)n_vars);
    uint64_t* d_scalars = (uint64_t*)c->d_ws;
    hipLaunchKernelGGL(srs_eq_scalars_kernel, dim3(mle_grid(n)), dim3(SRS_BLOCK), 0, c->stream, tau, n_vars, d_scalars);
    return srs_from_scalars(c, d_scalars, n, d_out_xy, d_out_inf);
}

extern "C" int zkhip_srs_univariate_g1(zkhip_ctx* c, const uint64_t* h_tau, size_t max_degree, uint64_t* d_out_xy,
                                       uint8_t* d_out_inf) {
    if (!c || !d_out_xy || !d_out_inf || !h_tau) return ZKHIP_ERR_ARG;
    ZK_TRY(c->activate());
    const size_t n = max_degree + 1;
    ZK_TRY(c->reserve_ws(((n * 32 + 255) & ~(size_t)255) + n * 192));
    FrArg tau = {};
    std::memcpy(tau.v, h_tau, 32);
    uint64_t* d_scalars = (uint64_t*)c->d_ws;
    hipLaunchKernelGGL(srs_power_scalars_kernel, dim3(mle_grid(n)), dim3(SRS_BLOCK), 0, c->stream, tau, n, d_scalars);
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


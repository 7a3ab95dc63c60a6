// msm_geometry.hpp -- the HOST side of the MSM's geometry: how every problem's 256 scalar bits are cut into digit windows and bucket
// sets, and the sizes of the passes that follow (msm.hip uploads the tables msm_build_geometry fills; the kernels of msm_kernels.hpp read
// them).  No HIP in here: the same code builds with g++ for the sanitizer run of tests/test_host_sanitizers_cpu.py.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "../../include/zkhip.h"

#ifdef __HIPCC__
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

constexpr int MSM_SEG_LOG = 3;              // L = 8 buckets per segment (L = 4 measured again with the row / column term sums: segment pass 0.30 instead of 0.33 ms, term sums 0.74 instead of 0.47 ms)
constexpr int MSM_SEG = 1 << MSM_SEG_LOG;
constexpr int MSM_MAX_WINS = 2048;          // digit windows of all problems of one pass: 64 problems x 32 windows of 8 bits, the narrowest a batch uses (the sort kernels keep the table in LDS: 32 KiB)

constexpr int SORT_MAX_PARTS = 4096;      // partitions of the counting sort's first level (msm_kernels.hpp)

// Geometry of one pass over one or several problems (built on the host: msm_build_geometry, msm.hip).  A problem's 256 scalar bits
// are cut into DIGIT WINDOWS of its own widths -- two widths one bit apart, so that they add up to exactly 256 and no window is sparse
// (a 255-bit scalar leaves the top window one spare bit: its digit never carries out) -- and every window owns a BUCKET SET of
// 2^(bits-1) buckets (bucket i holds digit magnitude i+1).  With the shifted-SRS table all windows of the one problem share one set.
struct MsmWin {
    uint32_t part_base;   // first sort partition of its bucket set
    uint32_t entry_off;   // shifted-SRS table: w * stride, added to the point index (0 otherwise)
    uint32_t bits;        // byte 0: window width c (digits in [-2^(c-1), 2^(c-1)]); byte 1: part_bits of its bucket set; byte 2: the low bits of a
    uint32_t pad;         //   bucket index that level 2 of the sort resolves (set width - 1 - part_bits; the set may be one bit wider than the window)
};
struct MsmSet {
    uint32_t bucket_base; // first bucket, a multiple of MSM_SEG
    uint32_t part_base;   // first sort partition: a partition holds the 2^(c-1-part_bits) <= 256 buckets with the same HIGH part_bits bits
    uint32_t bits;        //   of the index.  byte 0: c, byte 1: part_bits
    uint32_t term_base;   // its c - 3 (set, term) points start here
    uint32_t rc_base;     // its row / column sums (2 R + C points) start here
    uint32_t rcwg_base;   // its first workgroup in msm_rowcol_kernel's grid
    uint32_t termwg_base; // ... and in msm_rowcol_terms_kernel's
};
struct MsmPlan {
    uint32_t n_sets, n_buckets, n_parts, n_terms, n_rc, n_rcwg, n_termwg;
    uint32_t n_wins;      // digit windows in all
    uint32_t shared;      // 1: shifted-SRS table, one bucket set
    const MsmWin* wins;   // device tables
    const MsmSet* sets;
    const uint16_t* part_set;   // partition -> bucket set
    const uint16_t* rcwg_set;   // msm_rowcol_kernel workgroup -> bucket set
    const uint16_t* termwg_set; // msm_rowcol_terms_kernel workgroup -> bucket set
};
// Batched commits (several independent (points, scalars) problems laid end to end, e.g. the rounds of
// MultilinearKZG::open): problem j owns the entries [off[j], off[j+1]) and the digit windows [win_first[j], win_first[j+1]).
constexpr int MSM_MAX_PROBLEMS = 64;
struct MsmProblems {
    uint32_t n;
    uint32_t off[MSM_MAX_PROBLEMS + 1];
    uint16_t win_first[MSM_MAX_PROBLEMS + 1];
};
constexpr uint32_t MSM_LINE_Q = 4;
constexpr uint32_t MSM_LINE_MAX = 64 * MSM_LINE_Q;     // the longest line a wave sums: R = C = 256, the 2^19-bucket set of the shifted-SRS table
ZK_HD uint32_t msm_line_q(uint32_t count) { return count >= 4 * MSM_LINE_Q ? MSM_LINE_Q : 1; }
constexpr uint32_t MSM_TERMS_PER_WG = 4;     // (set, term) trees per wave of msm_rowcol_terms_kernel while a tree fits 16 lanes x 4 values
struct MsmSetShape { uint32_t n_bits, lo_bits, C, R, row_lines, col_lines, row_wgs, col_wgs, term_wgs; };
ZK_HD MsmSetShape msm_set_shape_c(uint32_t c) {
    MsmSetShape sh;
    sh.n_bits = c - 1 - MSM_SEG_LOG;
    sh.lo_bits = sh.n_bits / 2;
    sh.C = 1u << sh.lo_bits;
    sh.R = (1u << sh.n_bits) >> sh.lo_bits;
    sh.row_lines = sh.C <= 64 ? 64 / (sh.C / msm_line_q(sh.C)) : 1;          // rows (C values each) per wave; a longer line has a wave to itself
    sh.col_lines = sh.R <= 64 ? 64 / (sh.R / msm_line_q(sh.R)) : 1;          // columns (R values each) per wave
    sh.row_wgs = (2 * sh.R + sh.row_lines - 1) / sh.row_lines;   // the R rows of S, then the R rows of A
    sh.col_wgs = (sh.C + sh.col_lines - 1) / sh.col_lines;
    // (set, term) trees: <= max(R, C) values each; four per workgroup while a tree fits 16 lanes x 4 values, else one
    sh.term_wgs = (sh.R <= 64 && sh.C <= 64) ? (1 + sh.n_bits + MSM_TERMS_PER_WG - 1) / MSM_TERMS_PER_WG : 1 + sh.n_bits;
    return sh;
}

// ---- the geometry of a pass (see msm.hip for what it decides and the measurements behind the widths) ----------------------------------
struct MsmGeometry {
    MsmPlan pl = {};                 // totals; the device pointers are set by msm_enqueue
    std::vector<MsmWin> wins;
    std::vector<MsmSet> sets;
    std::vector<uint16_t> part_set, rcwg_set, termwg_set;
    std::vector<uint32_t> set_exp;   // weight 2^set_exp of a set's total (the first bit of its digit window; 0 with the table)
    std::vector<uint32_t> prob_set_first;   // problem j owns the sets [prob_set_first[j], prob_set_first[j+1])
    uint16_t win_first[MSM_MAX_PROBLEMS + 1] = {};
    size_t items = 0;                // (point, window) pairs of the pass
    uint32_t heavy_min = 32;
};
constexpr uint32_t MSM_TABLE_WINDOWS_MAX = 32;    // windows of the narrowest table (8 bits): the bound the size checks use
// LEVEL TABLES (the folded SRS levels of MultilinearKZG::open, zkhip_srs_level_tables): a batch of problems, each with a shifted table of
// its own -- problem j of n_j points has W_j = ceil(256 / c_j) windows of ~c_j = log2(n_j) bits (msm_level_table_widths), ONE bucket set of
// 2^(c_j - 1) ~ n_j / 2 buckets (as many as its ~17 sets of 2^(lg - 5) buckets without the table) and the entries
// T_j + w n_j + i = 2^(first bit of window w) * point i, T_j = sum_{k < j} W_k n_k: the tables of the batch's problems lie end to end in problem order.
// Widths: W = ceil(256 / lg) windows, the low n_hi of them `hi` bits wide and the rest hi - 1, exactly 256 bits in all -- no sparse top window,
// whose few distinct digits would pile a whole level into a handful of (heavy) buckets.
struct MsmLevelWidths { uint32_t W, hi, n_hi; };
inline MsmLevelWidths msm_level_table_widths(size_t nj, size_t batch_total) {
    int lg = 0;
    while (((size_t)1 << lg) < nj) ++lg;
    // measured (tools/sweep_level_delta.sh; ms at delta 0 / -1 / -2 / -3, plain batch in brackets): 2^15 2.04 / 1.61 / 1.46 / 1.30 [1.78],
    // 2^17 2.18 / 1.98 / 1.70 / 1.64 [1.91], 2^18 2.38 / 2.06 / 1.88 / 2.16 [2.23], 2^19 2.99 / 2.56 / 2.84 / - [3.25], 2^20 4.14 / - [4.97]:
    // a small opening is as long as its longest bucket lists (wider windows: more buckets, shorter lists), a large one is throughput bound
    int delta = batch_total < ((size_t)1 << 17) ? -3 : batch_total < ((size_t)1 << 18) ? -2 : batch_total < ((size_t)1 << 19) ? -1 : 0;
    // tuning aid: widest window log2(n_j) - delta.  Read ONCE per process: the layout of a table (W, hi, n_hi) is a function of the sizes
    // and of this value, and the table carries no header -- a value that changed between the build of a table and its use would
    // silently address it wrongly.  (Tables are not portable between processes that run with different values.)
    static const int env_delta = [] {
        const char* e = std::getenv("ZKHIP_LEVEL_TABLE_DELTA");
        if (!e) return 99;
        const int v = std::atoi(e);
        return v >= -3 && v <= 4 ? v : 99;
    }();
    if (env_delta != 99) delta = env_delta;
    const uint32_t c = (uint32_t)std::min(20, std::max(8, lg - delta));
    MsmLevelWidths lw;
    lw.W = (256 + c - 1) / c;
    lw.hi = (256 + lw.W - 1) / lw.W;
    lw.n_hi = 256 - lw.W * (lw.hi - 1);
    return lw;
}
// The shifted-SRS table of ONE commit over n_points points (zkhip_srs_precompute): the same rule -- thirteen windows of 20 / 19 bits on
// 2^19 buckets at 2^20 points, wider than the size below 2^19 points (a small commit is as long as its longest bucket list and its
// reduction passes: 2^19 buckets for 2^12 points were a millisecond of empty reduction).
inline MsmLevelWidths msm_table_widths(size_t n_points) { return msm_level_table_widths(n_points, n_points); }
inline int msm_build_geometry_at(const MsmProblems& pr, bool shared, size_t table_stride, int delta, MsmGeometry& g);
inline int msm_build_geometry(const MsmProblems& pr, bool shared, size_t table_stride, MsmGeometry& g);
inline int msm_build_geometry(const MsmProblems& pr, bool shared, size_t table_stride, MsmGeometry& g) {
    // widths log2(n_j) - delta, delta = the smallest from 1 on whose partitions fit the sort (MultilinearKZG::open: 3-4 at 2^20, where the
    // pass is throughput bound and the width hardly matters -- 4.92 / 4.97 / 5.36 ms at delta 4 / 3 / 5 -- and 1 below 2^19, where the accumulate
    // pass is as long as its longest lists: 2^16 1.66 / 1.74 / 1.96 / 2.77 ms at delta 1 / 2 / 3 / 4)
    int delta = 1;
    if (const char* e = std::getenv("ZKHIP_MSM_BATCH_DELTA")) {   // tuning aid (tools/perf_open.py): width = log2(n_j) - delta
        const int v = std::atoi(e);
        if (v >= 0 && v <= 8) delta = v;
    }
    int rc = ZKHIP_ERR_SHAPE;
    // narrower windows until the sort's partitions suffice (at 8 bits -- the floor -- 64 problems have 2048)
    for (; delta <= 24 && rc == ZKHIP_ERR_SHAPE; ++delta) rc = msm_build_geometry_at(pr, shared, table_stride, delta, g);
    return rc;
}
inline int msm_build_geometry_at(const MsmProblems& pr, bool shared, size_t table_stride, int delta, MsmGeometry& g) {
    g = MsmGeometry();
    uint32_t max_chain = 1, rc_max = 1;
    size_t level_entries = 0;
    g.prob_set_first.push_back(0);
    auto add_set = [&](uint32_t c, uint32_t exp) {
        MsmSet s = {};
        const uint32_t part_bits = c - 1 > 8 ? c - 1 - 8 : 0;
        s.bucket_base = g.pl.n_buckets;
        s.part_base = g.pl.n_parts;
        s.bits = c | (part_bits << 8);
        s.term_base = g.pl.n_terms;
        s.rc_base = g.pl.n_rc;
        s.rcwg_base = g.pl.n_rcwg;
        s.termwg_base = g.pl.n_termwg;
        const MsmSetShape sh = msm_set_shape_c(c);
        const uint32_t n_bits = sh.n_bits, C = sh.C, R = sh.R;
        const uint32_t idx = (uint32_t)g.sets.size();
        for (uint32_t p = 0; p < (1u << part_bits); ++p) g.part_set.push_back((uint16_t)idx);
        for (uint32_t b = 0; b < sh.row_wgs + sh.col_wgs; ++b) g.rcwg_set.push_back((uint16_t)idx);
        for (uint32_t t = 0; t < sh.term_wgs; ++t) g.termwg_set.push_back((uint16_t)idx);
        g.pl.n_buckets += 1u << (c - 1);
        g.pl.n_parts += 1u << part_bits;
        g.pl.n_terms += 1 + n_bits;
        g.pl.n_rc += 2 * R + C;
        g.pl.n_rcwg += sh.row_wgs + sh.col_wgs;
        g.pl.n_termwg += sh.term_wgs;
        rc_max = std::max(rc_max, std::max(R, C));
        g.sets.push_back(s);
        g.set_exp.push_back(exp);
        return s;
    };
    auto add_win = [&](const MsmSet& s, uint32_t c, uint32_t entry_off) {
        MsmWin w = {};
        w.part_base = s.part_base;
        w.entry_off = entry_off;
        const uint32_t set_c = s.bits & 0xffu, part_bits = (s.bits >> 8) & 0xffu;
        w.bits = c | (part_bits << 8) | ((set_c - 1 - part_bits) << 16);
        g.wins.push_back(w);
    };
    for (uint32_t j = 0; j < pr.n; ++j) {
        const size_t nj = pr.off[j + 1] - pr.off[j];
        g.win_first[j] = (uint16_t)g.wins.size();
        uint32_t lg = 0;
        while (((size_t)1 << lg) < nj) ++lg;
        if (shared && pr.n > 1) {                     // level tables: see msm_level_table_widths
            const MsmLevelWidths lw = msm_level_table_widths(nj, pr.off[pr.n]);
            const uint32_t c = lw.hi, W = lw.W;
            const MsmSet s = add_set(c, 0);
            if (level_entries + (size_t)W * nj >= ((size_t)1 << 31)) return ZKHIP_ERR_SHAPE;    // entry index + sign bit in 32 bits
            for (uint32_t w = 0; w < W; ++w)                                                        // + the point's index in the batch
                add_win(s, w < lw.n_hi ? lw.hi : lw.hi - 1, (uint32_t)(level_entries + (size_t)w * nj - pr.off[j]));
            level_entries += (size_t)W * nj;
            g.items += nj * W;
            max_chain = std::max<uint32_t>(max_chain, (uint32_t)((nj * W + (1u << (c - 1)) - 1) >> (c - 1)));
        } else if (shared) {                          // ONE problem against the shifted-SRS table of table_stride points (msm_table_widths)
            const MsmLevelWidths lw = msm_table_widths(table_stride);
            const MsmSet s = add_set(lw.hi, 0);
            for (uint32_t w = 0; w < lw.W; ++w) add_win(s, w < lw.n_hi ? lw.hi : lw.hi - 1, (uint32_t)(w * table_stride));
            g.items += nj * lw.W;
            max_chain = std::max<uint32_t>(max_chain, (uint32_t)((nj * lw.W + (1u << (lw.hi - 1)) - 1) >> (lw.hi - 1)));
        } else {
            uint32_t w, hi, n_hi;
            if (pr.n == 1) {
                uint32_t c = lg >= 13 ? 16 : lg >= 10 ? 12 : 8;
                if (const char* e = std::getenv("ZKHIP_MSM_C")) {   // tuning aid (tools/perf_msm.py); any 4 <= c <= 16 is correct
                    const int v = std::atoi(e);
                    if (v >= 4 && v <= 16) c = (uint32_t)v;
                }
                w = (256 + c - 1) / c; hi = c; n_hi = w;
            } else {
                const uint32_t c = (uint32_t)std::min(16, std::max(8, (int)lg - delta));   // >= 8 bits: at most 32 windows per problem, 2048 for 64 problems
                w = (256 + c - 1) / c;
                hi = (256 + w - 1) / w;
                n_hi = 256 - w * (hi - 1);      // n_hi windows of hi bits (the low ones), the rest of hi - 1: exactly 256 bits
            }
            uint32_t bit = 0;
            for (uint32_t v = 0; v < w; ++v) {
                const uint32_t c = v < n_hi ? hi : hi - 1;
                const MsmSet s = add_set(c, bit);
                add_win(s, c, 0);
                bit += c;
            }
            const uint32_t c_lo = n_hi < w ? hi - 1 : hi;
            g.items += nj * w;
            max_chain = std::max<uint32_t>(max_chain, (uint32_t)((nj + (1u << (c_lo - 1)) - 1) >> (c_lo - 1)));
        }
        g.prob_set_first.push_back((uint32_t)g.sets.size());
        if (g.wins.size() > (size_t)MSM_MAX_WINS) return ZKHIP_ERR_SHAPE;
    }
    g.win_first[pr.n] = (uint16_t)g.wins.size();
    g.pl.n_sets = (uint32_t)g.sets.size();
    g.pl.n_wins = (uint32_t)g.wins.size();
    g.pl.shared = shared ? 1u : 0u;
    if (g.pl.n_parts > (uint32_t)SORT_MAX_PARTS || g.pl.n_sets > 65535u) return ZKHIP_ERR_SHAPE;
    // a bucket holding more than heavy_min points (four average lists of the densest set) is summed by whole workgroups
    g.heavy_min = std::max<uint32_t>(32, 4 * max_chain);
    if (rc_max > MSM_LINE_MAX) return ZKHIP_ERR_SHAPE;      // a row / column is summed by one wave
    return ZKHIP_OK;
}


}  // namespace zk

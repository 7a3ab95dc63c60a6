// msm_kernels.hpp -- multi-scalar multiplication over BLS12-381 G1 for gfx950 (bucket method).
//
// Replaces the commit loops of the reference,
//     UnivariateKZG::commitment   kzg/src/univariate_kzg.rs:37-58
//     MultilinearKZG::commitment  kzg/src/multilinear_kzg.rs:33-48
// which compute sum_i srs[i].mul_bigint(coeff[i].into_bigint()) by double-and-add.  The group element is
// the same; the algorithm is Pippenger's:
//   1. digits:   scalars leave Montgomery form (into_bigint) and are recoded into signed digits, one per DIGIT WINDOW
//                (MsmWin: the widths are the geometry's, built on the host per problem);
//   2. sort:     the (point, window) pairs are counting-sorted by bucket in two levels without global atomics
//                (msm_sort_*), which also yields every bucket's count and offset;
//   3. order:    buckets are ranked by point count, heaviest first;
//   4. accumulate: one lane per bucket adds its points (mixed XYZZ additions); buckets far heavier than the
//                mean are filed by the sort and summed by whole workgroups in front of it (4b / 4c);
//   5. segments: every L consecutive buckets are folded by a local running sum into
//                S_s = sum B and A_s = sum (j+1) B  -- short dependency chains only;
//   6. terms:    per BUCKET SET (MsmSet), sum_s A_s and, for every bit k of the segment index, T_k = sum_{s: bit k} S_s
//                by row / column sums and trees inside single waves.  The set's total is sum_s A_s + L * sum_k 2^k T_k.
//   7. the (set, term) points, each tagged with its power-of-two weight, go to the host, which runs
//      the final 255-step double-and-add chain per problem (inherently serial; a few hundred group operations).
// One pass serves one problem or several independent ones (MsmProblems: problem j owns its own digit windows and bucket sets,
// each problem with window widths that suit its size -- all rounds of MultilinearKZG::open are one pass), or the shifted-SRS
// table (MsmPlan::shared: the points 2^(c w) P are read from a table built once per SRS, so the digits of all windows share
// ONE bucket set and one bucket reduction).
// MSM is integer-ALU bound (10 Fq products of ~500 instructions per added point, VALU issue saturated), not HBM bound.
#pragma once
#include "g1.hpp"
#include "g1u.hpp"
#include "msm_geometry.hpp"

namespace zk {

constexpr int MSM_BLOCK = 256;
__device__ __forceinline__ uint32_t msm_set_c(const MsmSet& s) { return s.bits & 0xffu; }
__device__ __forceinline__ uint32_t msm_set_part_bits(const MsmSet& s) { return (s.bits >> 8) & 0xffu; }

// Signed digit stream of a canonical scalar (8 x u32, little endian): next(c) takes the next c bits (1 <= c <= 31) and returns a
// digit in [-2^(c-1), 2^(c-1)].  The scalar is consumed by shifting (no dynamically indexed registers).
struct DigitStream {
    Fr v;
    uint32_t carry;
    __device__ __forceinline__ explicit DigitStream(const Fr& canon) : v(canon), carry(0) {}
    __device__ __forceinline__ int32_t next(uint32_t c) {
        uint32_t raw = (v.l[0] & ((1u << c) - 1)) + carry;
#pragma unroll
        for (int i = 0; i < 7; ++i) v.l[i] = (v.l[i] >> c) | (v.l[i + 1] << (32 - c));
        v.l[7] >>= c;
        if (raw > (1u << (c - 1))) { carry = 1; return (int32_t)raw - (int32_t)(1u << c); }
        carry = 0;
        return (int32_t)raw;
    }
};

// Bucket processing order: buckets sorted by their point count, heaviest first, so that the 64 lanes of a wave
// walk lists of (nearly) equal length.  Counting sort over the clamped count.
constexpr uint32_t MSM_COUNT_BINS = 1024;
// (the histogram itself is taken by msm_sort_local_kernel, which has every bucket's count in hand)
static __global__ __launch_bounds__(1024) void msm_order_scan_kernel(uint32_t* __restrict__ bins) {
    __shared__ uint32_t part[1024];
    const uint32_t v = bins[threadIdx.x];
    part[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    bins[threadIdx.x] = part[threadIdx.x] - v;
}
static __global__ __launch_bounds__(MSM_BLOCK) void msm_order_scatter_kernel(const uint32_t* __restrict__ counts, uint32_t n_buckets,
                                                                             uint32_t* __restrict__ bins,
                                                                             uint32_t* __restrict__ order) {
    __shared__ uint32_t local[MSM_COUNT_BINS];   // per-bin count of this workgroup, then its base position
    for (uint32_t i = threadIdx.x; i < MSM_COUNT_BINS; i += MSM_BLOCK) local[i] = 0;
    __syncthreads();
    const uint32_t b = blockIdx.x * MSM_BLOCK + threadIdx.x;
    uint32_t key = 0, rank = 0;
    if (b < n_buckets) {
        key = MSM_COUNT_BINS - 1 - min(counts[b], MSM_COUNT_BINS - 1);
        rank = atomicAdd(&local[key], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < MSM_COUNT_BINS; i += MSM_BLOCK)
        if (local[i]) local[i] = atomicAdd(&bins[i], local[i]);
    __syncthreads();
    if (b < n_buckets) order[local[key] + rank] = b;
}

// ---- counting sort in two levels, without global atomics -------------------------------------------------
// Level 1 splits the (point, window) pairs by (window, high bits of the bucket) into <= 4096 partitions; a workgroup
// counts its tile in LDS (sort_count), a column scan turns the per-workgroup counts into exclusive write positions
// (sort_bases), and the same tile walk scatters 8-byte items (sort_scatter).  Level 2 runs one workgroup per
// partition: the remaining <= 8 bucket bits are resolved in LDS, which also yields every bucket's count and offset.
constexpr int SORT_TILE = 2048;          // scalars per workgroup tile (8 per lane; 1024: count -20 us, bases +24; 4096: count and scatter +40 us each)

// the problems' ranges and the digit windows in LDS (read per scalar / per (scalar, window) by both passes over the scalars; the
// kernel argument itself, indexed by a value that differs from lane to lane, is read through a loop over the lanes' values)
struct MsmSortTables {
    uint4 wl[MSM_MAX_WINS + 1];          // one entry of slack: the loops read the descriptor behind the last window
    uint32_t off[MSM_MAX_PROBLEMS + 1];
    uint32_t win_first[MSM_MAX_PROBLEMS + 1];
};
__device__ __forceinline__ void msm_load_tables(const MsmPlan& pl, const MsmProblems& pr, MsmSortTables& t) {
    const uint4* src = reinterpret_cast<const uint4*>(pl.wins);
    for (uint32_t i = threadIdx.x; i < pl.n_wins; i += MSM_BLOCK) t.wl[i] = src[i];
    for (uint32_t i = threadIdx.x; i <= pr.n; i += MSM_BLOCK) { t.off[i] = pr.off[i]; t.win_first[i] = pr.win_first[i]; }
}
__device__ __forceinline__ uint32_t msm_problem_of(const MsmSortTables& t, uint32_t n_problems, uint32_t i) {
    uint32_t lo = 0, hi = n_problems;      // off[lo] <= i < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (i >= t.off[mid]) lo = mid; else hi = mid;
    }
    return lo;
}
// the partition of a digit: the HIGH part_bits bits of its bucket index (all windows of every geometry are dense, so the partitions of a
// set fill evenly; level 2 resolves the low <= 8 bits and writes a partition's counts / offsets / sorted run contiguously)
__device__ __forceinline__ uint32_t msm_win_sub_bits(const uint4& w) { return (w.z >> 16) & 0xffu; }   // w = an MsmWin
__device__ __forceinline__ uint32_t msm_partition_of(const uint4& w, uint32_t mag) { return w.x + ((mag - 1) >> msm_win_sub_bits(w)); }

static __global__ __launch_bounds__(MSM_BLOCK) void msm_sort_count_kernel(const uint64_t* __restrict__ scalars,
                                                                          const uint8_t* __restrict__ inf, size_t n, MsmPlan pl,
                                                                          MsmProblems pr, uint32_t* __restrict__ wg_counts) {
    __shared__ uint32_t local[SORT_MAX_PARTS];
    __shared__ MsmSortTables tab;
    for (uint32_t i = threadIdx.x; i < pl.n_parts; i += MSM_BLOCK) local[i] = 0;
    msm_load_tables(pl, pr, tab);
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    for (uint32_t u = 0; u < SORT_TILE / MSM_BLOCK; ++u) {
        const size_t i = base + u * MSM_BLOCK + threadIdx.x;
        if (i >= n || (inf && inf[i])) continue;
        DigitStream ds(load_fr(scalars, i).from_mont());
        const uint32_t j = pr.n > 1 ? msm_problem_of(tab, pr.n, (uint32_t)i) : 0u;
        uint32_t v = tab.win_first[j];
        const uint32_t v_end = tab.win_first[j + 1];
        uint4 w = tab.wl[v];
        for (; v < v_end; ++v) {
            const uint4 w_next = tab.wl[v + 1];      // the next window's descriptor is on its way while this digit is cut (one entry of slack behind the table)
            const int32_t d = ds.next(w.z & 0xffu);
            if (d != 0) atomicAdd(&local[msm_partition_of(w, d < 0 ? (uint32_t)(-d) : (uint32_t)d)], 1u);
            w = w_next;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < pl.n_parts; i += MSM_BLOCK) wg_counts[(size_t)blockIdx.x * pl.n_parts + i] = local[i];
}

// per partition: exclusive scan of the workgroups' counts (in place) and the partition total.  One WAVE per partition:
// lane l owns the workgroups [l * per, (l + 1) * per), the lanes' sums are scanned with shuffles (a single lane walking
// all the workgroups of a partition is a chain of dependent strided loads: 120 us at 2^20 points).
static __global__ __launch_bounds__(MSM_BLOCK) void msm_sort_bases_kernel(uint32_t* __restrict__ wg_counts, uint32_t n_wgs,
                                                                          uint32_t n_parts, uint32_t* __restrict__ part_count) {
    const uint32_t p = blockIdx.x * (MSM_BLOCK / 64) + (threadIdx.x >> 6);
    if (p >= n_parts) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t per = (n_wgs + 63) / 64;
    const uint32_t lo = lane * per, hi = min(lo + per, n_wgs);
    uint32_t sum = 0;
    for (uint32_t g = lo; g < hi; ++g) sum += wg_counts[(size_t)g * n_parts + p];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += v;
    }
    uint32_t run = incl - sum;
    for (uint32_t g = lo; g < hi; ++g) {
        const uint32_t c = wg_counts[(size_t)g * n_parts + p];
        wg_counts[(size_t)g * n_parts + p] = run;
        run += c;
    }
    if (lane == 63) part_count[p] = incl;
}
// exclusive scan of <= SORT_MAX_PARTS partition totals by one workgroup (four consecutive ones per lane); part_off[n_parts] = grand total
static __global__ __launch_bounds__(1024) void msm_sort_part_scan_kernel(const uint32_t* __restrict__ part_count, uint32_t n_parts,
                                                                         uint32_t* __restrict__ part_off) {
    static_assert(SORT_MAX_PARTS <= 4 * 1024, "four partitions per lane");
    __shared__ uint32_t part[1024];
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t a = 4 * threadIdx.x + u;
        v[u] = a < n_parts ? part_count[a] : 0;
        sum += v[u];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t t = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t a = 4 * threadIdx.x + u;
        if (a < n_parts) part_off[a] = run;
        run += v[u];
    }
    if (threadIdx.x == 1023) part_off[n_parts] = part[1023];
}

static __global__ __launch_bounds__(MSM_BLOCK) void msm_sort_scatter_kernel(const uint64_t* __restrict__ scalars,
                                                                            const uint8_t* __restrict__ inf, size_t n, MsmPlan pl,
                                                                            MsmProblems pr, const uint32_t* __restrict__ wg_bases,
                                                                            const uint32_t* __restrict__ part_off,
                                                                            uint2* __restrict__ items) {
    __shared__ uint32_t cur[SORT_MAX_PARTS];
    __shared__ MsmSortTables tab;
    for (uint32_t i = threadIdx.x; i < pl.n_parts; i += MSM_BLOCK)
        cur[i] = part_off[i] + wg_bases[(size_t)blockIdx.x * pl.n_parts + i];
    msm_load_tables(pl, pr, tab);
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    for (uint32_t u = 0; u < SORT_TILE / MSM_BLOCK; ++u) {
        const size_t i = base + u * MSM_BLOCK + threadIdx.x;
        if (i >= n || (inf && inf[i])) continue;
        DigitStream ds(load_fr(scalars, i).from_mont());
        const uint32_t j = pr.n > 1 ? msm_problem_of(tab, pr.n, (uint32_t)i) : 0u;
        uint32_t v = tab.win_first[j];
        const uint32_t v_end = tab.win_first[j + 1];
        uint4 w = tab.wl[v];
        for (; v < v_end; ++v) {
            const uint4 w_next = tab.wl[v + 1];
            const int32_t d = ds.next(w.z & 0xffu);
            if (d != 0) {
                const bool neg = d < 0;
                const uint32_t mag = neg ? (uint32_t)(-d) : (uint32_t)d;
                const uint32_t pos = atomicAdd(&cur[msm_partition_of(w, mag)], 1u);
                items[pos] = make_uint2((w.y + (uint32_t)i) | (neg ? 0x80000000u : 0u), (mag - 1) & ((1u << msm_win_sub_bits(w)) - 1));
            }
            w = w_next;
        }
    }
}

// Heavy buckets (pass 4 below walks a bucket with one lane): a bucket holding more than `heavy_min` points is filed as records of
// <= 2048 points for pass 4b, where a whole workgroup sums a record (<= 8 points per lane, then a tree in LDS), and, when the bucket
// spans several records, as a tree of <= 256-way sums over the record sums for pass 4c.  With uniform scalars no bucket is heavy.
constexpr uint32_t MSM_HEAVY_LANE_MAX = 8;                        // points per lane inside a record (32 at first: the pass is a latency chain of that many additions plus the tree)
constexpr uint32_t MSM_HEAVY_REC = MSM_BLOCK * MSM_HEAVY_LANE_MAX;   // points per record
constexpr int MSM_HEAVY_LEVELS = 4;                               // 0: records of points; 1..3: 256-way sums of sums
struct MsmHeavyRec {
    uint32_t first;    // level 0: position in `sorted`; level >= 1: slot in the partial-sum array
    uint32_t count;    // points / partial sums covered
    uint32_t dst;      // slot of the result, or the bucket index when `final`
    uint32_t final;
};
struct MsmOverflow {
    uint32_t n_slots;
    uint32_t n_rec[MSM_HEAVY_LEVELS];
};

// files bucket b (cnt points from `start`) into the record lists; rec[l] has room for rec_cap records per level
__device__ __noinline__ void msm_file_heavy(uint32_t b, uint32_t start, uint32_t cnt, MsmOverflow* __restrict__ ovf,
                                            MsmHeavyRec* __restrict__ rec, uint32_t rec_cap) {
    uint32_t groups = (cnt + MSM_HEAVY_REC - 1) / MSM_HEAVY_REC;
    MsmHeavyRec r;
    if (groups == 1) {
        r.first = start; r.count = cnt; r.dst = b; r.final = 1;
        rec[atomicAdd(&ovf->n_rec[0], 1u)] = r;
        return;
    }
    uint32_t slot = atomicAdd(&ovf->n_slots, groups);
    uint32_t at = atomicAdd(&ovf->n_rec[0], groups);
    for (uint32_t g = 0; g < groups; ++g) {
        r.first = start + g * MSM_HEAVY_REC; r.count = min(MSM_HEAVY_REC, cnt - g * MSM_HEAVY_REC); r.dst = slot + g; r.final = 0;
        rec[at + g] = r;
    }
    uint32_t first = slot, count = groups;
    for (int level = 1; level < MSM_HEAVY_LEVELS; ++level) {
        MsmHeavyRec* lst = rec + (size_t)level * rec_cap;
        groups = (count + MSM_BLOCK - 1) / MSM_BLOCK;
        if (groups == 1) {
            r.first = first; r.count = count; r.dst = b; r.final = 1;
            lst[atomicAdd(&ovf->n_rec[level], 1u)] = r;
            return;
        }
        slot = atomicAdd(&ovf->n_slots, groups);
        at = atomicAdd(&ovf->n_rec[level], groups);
        for (uint32_t g = 0; g < groups; ++g) {
            r.first = first + g * MSM_BLOCK; r.count = min((uint32_t)MSM_BLOCK, count - g * MSM_BLOCK); r.dst = slot + g; r.final = 0;
            lst[at + g] = r;
        }
        first = slot; count = groups;
    }
}

// level 2: one workgroup per partition; writes the final order plus counts / offsets of its 2^sub_bits buckets.
// Skewed scalars make partitions of very different sizes (all points of a 0/1 table fall into one), so the
// workgroup is wide (1024 lanes of loads in flight) and a wave whose lanes all hold the same bucket -- the case that
// would serialise 64-fold on one LDS word -- issues a single aggregated atomic.
constexpr int SORT_LOCAL_BLOCK = 1024;
__device__ __forceinline__ uint32_t msm_lds_rank(uint32_t* bins, uint32_t key) {   // atomicAdd(&bins[key], 1), aggregated
    const uint64_t active = __ballot(1);
    const uint32_t first = __builtin_amdgcn_readfirstlane(key);
    if (__ballot(key == first) == active) {
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t below = __popcll(active & (((uint64_t)1 << lane) - 1));
        uint32_t base = 0;
        if (below == 0) base = atomicAdd(&bins[first], (uint32_t)__popcll(active));
        return __builtin_amdgcn_readfirstlane(base) + below;
    }
    return atomicAdd(&bins[key], 1u);
}
static __global__ __launch_bounds__(SORT_LOCAL_BLOCK) void msm_sort_local_kernel(const uint2* __restrict__ items,
                                                                                 const uint32_t* __restrict__ part_off, MsmPlan pl,
                                                                                 uint32_t* __restrict__ sorted,
                                                                                 uint32_t* __restrict__ counts,
                                                                                 uint32_t* __restrict__ offsets, uint32_t heavy_min,
                                                                                 MsmOverflow* __restrict__ ovf, MsmHeavyRec* __restrict__ rec,
                                                                                 uint32_t rec_cap, uint32_t* __restrict__ count_bins) {
    __shared__ uint32_t bins[256];
    __shared__ uint32_t scan[256];
    __shared__ uint32_t hist[MSM_COUNT_BINS];     // this partition's share of the bucket-count histogram (pass 3 orders the buckets by it)
    static_assert(SORT_LOCAL_BLOCK == (int)MSM_COUNT_BINS, "one histogram bin per lane");
    hist[threadIdx.x] = 0;
    const uint32_t p = blockIdx.x;
    const uint32_t lo = part_off[p], hi = part_off[p + 1];
    const MsmSet set = pl.sets[pl.part_set[p]];
    const uint32_t sub_bits = msm_set_c(set) - 1 - msm_set_part_bits(set);
    const uint32_t n_sub = 1u << sub_bits;                              // the partition's buckets: the low bits of the index (<= 8)
    if (threadIdx.x < 256) bins[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t q = lo + threadIdx.x; q < hi; q += SORT_LOCAL_BLOCK) msm_lds_rank(bins, items[q].y);
    __syncthreads();
    // exclusive scan of the (<= 256) bins
    const uint32_t v = threadIdx.x < n_sub ? bins[threadIdx.x] : 0;
    if (threadIdx.x < 256) scan[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t t = (threadIdx.x < 256 && threadIdx.x >= d) ? scan[threadIdx.x - d] : 0;
        __syncthreads();
        if (threadIdx.x < 256) scan[threadIdx.x] += t;
        __syncthreads();
    }
    if (threadIdx.x < n_sub) {
        const uint32_t excl = scan[threadIdx.x] - v;
        const uint32_t bucket = set.bucket_base + ((p - set.part_base) << sub_bits) + threadIdx.x;
        counts[bucket] = v;
        offsets[bucket] = lo + excl;
        bins[threadIdx.x] = lo + excl;      // becomes the write cursor
        atomicAdd(&hist[MSM_COUNT_BINS - 1 - min(v, MSM_COUNT_BINS - 1)], 1u);
        if (v > heavy_min) msm_file_heavy(bucket, lo + excl, v, ovf, rec, rec_cap);     // passes 4b / 4c sum it
    }
    __syncthreads();
    if (hist[threadIdx.x]) atomicAdd(&count_bins[threadIdx.x], hist[threadIdx.x]);     // counts cluster around their mean: few, hot bins
    for (uint32_t q = lo + threadIdx.x; q < hi; q += SORT_LOCAL_BLOCK) {
        const uint2 it = items[q];
        sorted[msm_lds_rank(bins, it.y)] = it.x;
    }
}

// pass 0: SRS points from the arkworks layout (96 B) into the internal unsaturated layout (128 B), once per commit
static __global__ __launch_bounds__(MSM_BLOCK) void msm_convert_points_kernel(const uint64_t* __restrict__ points, size_t n,
                                                                              uint32_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * MSM_BLOCK;
    for (size_t i = (size_t)blockIdx.x * MSM_BLOCK + threadIdx.x; i < n; i += stride) {
        G1Affine a = load_affine(points, i);
        store_fqu(out + 32 * i, fqu_from_ark(a.x));
        store_fqu(out + 32 * i + 16, fqu_from_ark(a.y));
    }
}

// pass 4: one lane per bucket (in the order given by `order`: heaviest buckets first, equal lengths inside a wave).
// Skewed scalars (small table values, constant polynomials, a sparse top window) can put a large share of all
// points into a handful of buckets, and one lane adding 2^20 points one after the other would take seconds: the
// heavy buckets (filed by the sort, above) are left to passes 4b / 4c.
__device__ __forceinline__ G1XyzzU msm_sum_run(const uint32_t* __restrict__ points, const uint32_t* __restrict__ sorted,
                                               uint32_t start, uint32_t cnt) {
    G1XyzzU acc = G1XyzzU::identity();
    // software pipeline: the next point's index and coordinates (a dependent pair of random loads) are in flight
    // while the current addition (~6 k instructions) runs
    uint32_t e_next = cnt ? sorted[start] : 0u;
    G1AffineU p_next = load_affine_u(points, e_next & 0x7fffffffu);
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t e = e_next;
        const G1AffineU p = p_next;
        if (k + 1 < cnt) {
            e_next = sorted[start + k + 1];
            p_next = load_affine_u(points, e_next & 0x7fffffffu);
        }
        g1u_madd(acc, p, (e >> 31) != 0);
    }
    return acc;
}

static __global__ __launch_bounds__(MSM_BLOCK) void msm_accumulate_kernel(const uint32_t* __restrict__ points,
                                                                   const uint32_t* __restrict__ sorted,
                                                                   const uint32_t* __restrict__ offsets,
                                                                   const uint32_t* __restrict__ counts,
                                                                   const uint32_t* __restrict__ order,
                                                                   uint32_t n_buckets, uint32_t heavy_min, uint32_t* __restrict__ buckets) {
    const uint32_t t = blockIdx.x * MSM_BLOCK + threadIdx.x;
    if (t >= n_buckets) return;
    const uint32_t b = order[t];
    const uint32_t start = offsets[b], cnt = counts[b];
    if (cnt > heavy_min) return;                            // filed by msm_sort_local_kernel; passes 4b / 4c write buckets[b]
    store_xyzz_u(buckets, b, msm_sum_run(points, sorted, start, cnt));
}
// tree sum over the first `width` lanes' values (width a power of two <= MSM_BLOCK); the result is lane 0's `acc`
__device__ __forceinline__ void msm_block_tree_sum(G1XyzzU& acc, uint32_t width, uint32_t* __restrict__ lds) {
    if (threadIdx.x < width) store_xyzz_u(lds, threadIdx.x, acc);
    __syncthreads();
    for (uint32_t d = width >> 1; d >= 1; d >>= 1) {
        if (threadIdx.x < d) {
            G1XyzzU o = load_xyzz_u(lds, threadIdx.x + d);
            g1u_add(acc, o);
            store_xyzz_u(lds, threadIdx.x, acc);
        }
        __syncthreads();
    }
}

// the same inside ONE wave without LDS or barriers: aligned groups of `width` lanes (a power of two <= 64) are summed independently,
// the result of a group is in its first lane.  The operands travel by ds_bpermute (56 words per level, against ~7 k instructions per addition).
__device__ __forceinline__ G1XyzzU msm_shfl_down(const G1XyzzU& v, uint32_t d) {
    G1XyzzU o;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) {
        o.x.l[i] = __shfl_down(v.x.l[i], d, 64);
        o.y.l[i] = __shfl_down(v.y.l[i], d, 64);
        o.zz.l[i] = __shfl_down(v.zz.l[i], d, 64);
        o.zzz.l[i] = __shfl_down(v.zzz.l[i], d, 64);
    }
    return o;
}
__device__ __forceinline__ G1XyzzU msm_shfl(const G1XyzzU& v, uint32_t src_lane) {
    G1XyzzU o;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) {
        o.x.l[i] = __shfl(v.x.l[i], src_lane, 64);
        o.y.l[i] = __shfl(v.y.l[i], src_lane, 64);
        o.zz.l[i] = __shfl(v.zz.l[i], src_lane, 64);
        o.zzz.l[i] = __shfl(v.zzz.l[i], src_lane, 64);
    }
    return o;
}
__device__ __forceinline__ void msm_wave_tree_sum_lanes(G1XyzzU& acc, uint32_t width) {      // a lane per pair: one full addition per level
    const uint32_t pos = threadIdx.x & (width - 1);
    for (uint32_t d = width >> 1; d >= 1; d >>= 1) {
        G1XyzzU o = msm_shfl_down(acc, d);
        if (pos < d) g1u_add(acc, o);
    }
}
// The same with a QUAD of lanes per pair (g1u_add_quad: an addition in ~2.6 k instructions instead of ~7 k): a level of P pairs is
// ceil(P / 16) passes -- the pairs' operands gathered onto their quads by ds_bpermute, the sums scattered back to the pairs' first
// lanes.  A 64-lane tree: 2 + 1 + 1 + 1 + 1 + 1 passes instead of six full additions.  The wave must be 64 lanes, all of them calling.
__device__ __forceinline__ void msm_wave_tree_sum(G1XyzzU& acc, uint32_t width) {
    const uint32_t lane = threadIdx.x & 63, quad = lane >> 2, pos = lane & (width - 1), line = lane / width, lines = 64 / width;
    for (uint32_t d = width >> 1, ld = 31 - __builtin_clz(width >> 1); d >= 1; d >>= 1, --ld) {
        const uint32_t pairs = lines * d;                  // pair t = (line, pos < d): lanes line * width + pos and ... + d
        const uint32_t my_t = line * d + pos;              // the pair this lane is the first lane of (when pos < d)
        for (uint32_t t0 = 0; t0 < pairs; t0 += 16) {
            const uint32_t t = t0 + quad;
            const bool live = t < pairs;
            const uint32_t la = live ? (t >> ld) * width + (t & (d - 1)) : lane;
            G1XyzzU a = msm_shfl(acc, la), b = msm_shfl(acc, live ? la + d : lane);
            if (!live) b = G1XyzzU::identity();            // an idle quad: a + 0, nobody reads it
            const G1XyzzU s = g1u_add_quad(a, b);
            const bool mine = pos < d && my_t >= t0 && my_t < t0 + 16;
            const G1XyzzU got = msm_shfl(s, mine ? 4 * (my_t - t0) : lane);
            if (mine) acc = got;
        }
        if (d == 1) break;
    }
}

// pass 4b: one workgroup per level-0 record (fixed grid; the list length is read on the device)
static __global__ __launch_bounds__(MSM_BLOCK) void msm_heavy_points_kernel(const uint32_t* __restrict__ points,
                                                                     const uint32_t* __restrict__ sorted,
                                                                     const MsmOverflow* __restrict__ ovf,
                                                                     const MsmHeavyRec* __restrict__ rec,
                                                                     uint32_t* __restrict__ partials,
                                                                     uint32_t* __restrict__ buckets) {
    __builtin_amdgcn_s_setprio(3);   // a latency-bound pass: beside another commit's accumulate pass its few waves win the issue arbitration
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(zk_dyn_lds);   // MSM_BLOCK x 64 u32
    const uint32_t n_rec = ovf->n_rec[0];
    for (uint32_t i = blockIdx.x; i < n_rec; i += gridDim.x) {
        const MsmHeavyRec r = rec[i];
        // 8 points per lane (the tree's general additions stay a small share of the work); the last lanes of a short record fewer
        const uint32_t per = max(8u, (r.count + MSM_BLOCK - 1) / MSM_BLOCK);
        const uint32_t lanes = (r.count + per - 1) / per;
        uint32_t width = 1;
        while (width < lanes) width <<= 1;
        G1XyzzU acc = G1XyzzU::identity();
        if (threadIdx.x < lanes) {
            const uint32_t lo = threadIdx.x * per;
            acc = msm_sum_run(points, sorted, r.first + lo, min(per, r.count - lo));
        }
        msm_block_tree_sum(acc, width, lds);
        if (threadIdx.x == 0) store_xyzz_u(r.final ? buckets : partials, r.dst, acc);
    }
}

// pass 4c (one launch per level >= 1): one workgroup per record sums <= 256 partial sums
static __global__ __launch_bounds__(MSM_BLOCK) void msm_heavy_tree_kernel(const MsmOverflow* __restrict__ ovf, uint32_t level,
                                                                   const MsmHeavyRec* __restrict__ rec,
                                                                   uint32_t* __restrict__ partials,
                                                                   uint32_t* __restrict__ buckets) {
    __builtin_amdgcn_s_setprio(3);   // a latency-bound pass: beside another commit's accumulate pass its few waves win the issue arbitration
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(zk_dyn_lds);
    const uint32_t n_rec = ovf->n_rec[level];
    for (uint32_t i = blockIdx.x; i < n_rec; i += gridDim.x) {
        const MsmHeavyRec r = rec[i];
        uint32_t width = 1;
        while (width < r.count) width <<= 1;
        G1XyzzU acc = G1XyzzU::identity();
        if (threadIdx.x < r.count) acc = load_xyzz_u(partials, r.first + threadIdx.x);
        msm_block_tree_sum(acc, width, lds);
        if (threadIdx.x == 0) store_xyzz_u(r.final ? buckets : partials, r.dst, acc);
    }
}

// pass 5: one lane per segment of L buckets: S = sum_j B_j, A = sum_j (j+1) B_j  (running sum from the top)
static __global__ __launch_bounds__(MSM_BLOCK) void msm_segment_kernel(const uint32_t* __restrict__ buckets,
                                                                uint32_t n_segments, uint32_t* __restrict__ seg_s,
                                                                uint32_t* __restrict__ seg_a) {
    __builtin_amdgcn_s_setprio(2);     // latency-bound chain: beside another commit's accumulate pass (commits in flight) it must win the issue slots
    const uint32_t s = blockIdx.x * MSM_BLOCK + threadIdx.x;
    if (s >= n_segments) return;
    G1XyzzU running = G1XyzzU::identity(), acc = G1XyzzU::identity();
    for (int j = MSM_SEG - 1; j >= 0; --j) {
        G1XyzzU bkt = load_xyzz_u(buckets, (size_t)s * MSM_SEG + j);
        g1u_add(running, bkt);
        g1u_add(acc, running);
    }
    store_xyzz_u(seg_s, s, running);
    store_xyzz_u(seg_a, s, acc);
}

// pass 6, per bucket set (ns = 2^n_bits segments, n_bits = c - 1 - log2 L): the set's total is sum_s A_s + L sum_k 2^k T_k with
// T_k = sum_{s: bit k} S_s.  By rows and columns: write the segment index as s = h C + l (C = 2^lo_bits columns, R = ns / C rows).  Then
//     T_k = sum_{l: bit k} colS_l            (k < lo_bits),   colS_l = sum_h S_{h C + l}
//         = sum_{h: bit k - lo_bits} rowS_h  (k >= lo_bits),  rowS_h = sum_l S_{h C + l}
// and sum_s A_s = sum_h rowA_h: 3 ns additions in R + R + C independent trees, then 1 + n_bits trees over <= max(R, C)
// values -- instead of ns (1 + n_bits / 2) additions in chains of a dozen and trees of 256 (the shared bucket set of the
// table path: 2^16 segments, 0.56 ms of mostly idle chip).  Every addition is ~7 k instructions (~15 us on a lone wave), so
// what counts is the depth: log2 C + log2 R tree levels.
// A tree of <= 64 values is summed inside one wave (msm_wave_tree_sum: no LDS, so the waves of a commit are all resident at once --
// with a 16 KiB tree buffer per workgroup ten fitted a CU and the 3072 trees of a 2^20-point commit ran in two rounds), and a wave
// takes as many lines (rows / columns) of its set as fit; the 256-value lines of the table path's one wide set take a wave each
// (64 lanes x 4 values).  The grid is flat over the sets, which differ in size: rcwg_set names the set of a workgroup.
// A lane first adds MSM_LINE_Q values of its line one after the other, then the lanes of the line form the tree: a tree level costs a
// whole wave one addition however few lanes still take part, so a 64-value line as 16 lanes x 4 values is 3 + 4 additions for FOUR lines
// per wave instead of 6 for one (the row / column pass of a 2^20-point commit: 768 waves, each alone on a SIMD, instead of 3072).
static __global__ __launch_bounds__(64) void msm_rowcol_kernel(const uint32_t* __restrict__ seg_s, const uint32_t* __restrict__ seg_a,
                                                             MsmPlan pl, uint32_t* __restrict__ rc) {
    __builtin_amdgcn_s_setprio(2);
    const MsmSet set = pl.sets[pl.rcwg_set[blockIdx.x]];
    const MsmSetShape sh = msm_set_shape_c(msm_set_c(set));
    const uint32_t C = sh.C, R = sh.R, b = blockIdx.x - set.rcwg_base;
    const size_t seg0 = set.bucket_base >> MSM_SEG_LOG;
    uint32_t* row_s = rc + (size_t)set.rc_base * 64;    // R row sums of S, R of A, C column sums of S
    uint32_t* row_a = row_s + (size_t)R * 64;
    uint32_t* col_s = row_a + (size_t)R * 64;
    const bool rows = b < sh.row_wgs;
    const uint32_t count = rows ? C : R;                 // values per line (<= MSM_LINE_MAX)
    // lane = (line of this wave, position in the line), q values per lane one after the other, then the tree over the line's lanes
    const uint32_t lpl = count <= 64 ? count / msm_line_q(count) : 64;      // lanes per line
    const uint32_t q = count / lpl;
    const uint32_t lines = rows ? sh.row_lines : sh.col_lines;
    const uint32_t line = (rows ? b : b - sh.row_wgs) * lines + threadIdx.x / lpl, pos = threadIdx.x & (lpl - 1);
    const uint32_t n_lines = rows ? 2 * R : C;
    G1XyzzU acc = G1XyzzU::identity();
    uint32_t* dst = nullptr;
    if (line < n_lines) {
        const uint32_t* src;
        size_t first, stride;
        if (rows) {
            const uint32_t h = line < R ? line : line - R;
            src = (line < R ? seg_s : seg_a) + seg0 * 64; first = (size_t)h * C + (size_t)pos * q; stride = 1;
            dst = (line < R ? row_s : row_a) + (size_t)h * 64;
        } else {
            src = seg_s + seg0 * 64; first = (size_t)pos * q * C + line; stride = C;
            dst = col_s + (size_t)line * 64;
        }
        acc = load_xyzz_u(src, first);
        for (uint32_t u = 1; u < q; ++u) {
            G1XyzzU v = load_xyzz_u(src, first + u * stride);
            g1u_add(acc, v);
        }
    }
    msm_wave_tree_sum(acc, lpl);
    if (dst && pos == 0) store_xyzz_u(dst, 0, acc);
}
// ... then one tree per (set, term): term 0 = sum_s A_s, term 1 + k = T_k; the result leaves in the arkworks layout (XYZZ, 4 x 48 B)
// for the host epilogue.  Four trees of a set per wave (16 lanes x <= 4 selected values each) while R, C <= 64; a wave per tree (64 lanes
// x <= 4 values) for the wide set of the table path.  termwg_set names the set of a workgroup.
static __global__ __launch_bounds__(64) void msm_rowcol_terms_kernel(const uint32_t* __restrict__ rc, MsmPlan pl, uint64_t* __restrict__ terms) {
    __builtin_amdgcn_s_setprio(2);
    const MsmSet set = pl.sets[pl.termwg_set[blockIdx.x]];
    const MsmSetShape sh = msm_set_shape_c(msm_set_c(set));
    const uint32_t C = sh.C, R = sh.R, lo_bits = sh.lo_bits;
    const bool packed = R <= 64 && C <= 64;
    const uint32_t lanes = packed ? 64 / MSM_TERMS_PER_WG : 64;         // lanes of one tree
    const uint32_t t = packed ? (blockIdx.x - set.termwg_base) * MSM_TERMS_PER_WG + threadIdx.x / lanes : blockIdx.x - set.termwg_base;
    const uint32_t pos = threadIdx.x & (lanes - 1);
    const bool live = t < 1 + sh.n_bits;
    const uint32_t* row_s = rc + (size_t)set.rc_base * 64;
    const uint32_t* row_a = row_s + (size_t)R * 64;
    const uint32_t* col_s = row_a + (size_t)R * 64;
    const uint32_t* src;
    uint32_t count, bit = 0;
    const bool select = t != 0;
    if (t == 0) { src = row_a; count = R; }
    else if (t - 1 < lo_bits) { src = col_s; count = C; bit = t - 1; }
    else { src = row_s; count = R; bit = t - 1 - lo_bits; }
    G1XyzzU acc = G1XyzzU::identity();
    // selected entries: every one (term 0), or those with `bit` set -- the j-th of them is j with a one inserted at `bit`
    const uint32_t n_sel = live ? (select ? count >> 1 : count) : 0;
    for (uint32_t j = pos; j < n_sel; j += lanes) {
        uint32_t i = j;
        if (select) i = ((j >> bit) << (bit + 1)) | (1u << bit) | (j & ((1u << bit) - 1));
        G1XyzzU v = load_xyzz_u(src, i);
        g1u_add(acc, v);
    }
    msm_wave_tree_sum(acc, lanes);
    if (live && pos == 0) {
        uint64_t* o = terms + 24 * ((size_t)set.term_base + t);
        store_fq(o, fqu_to_ark(acc.x));
        store_fq(o + 6, fqu_to_ark(acc.y));
        store_fq(o + 12, fqu_to_ark(acc.zz));
        store_fq(o + 18, fqu_to_ark(acc.zzz));
    }
}

// ---- commits of a few thousand points against the shifted-SRS table: no sort, no buckets ------------------------------------------
// The bucket pipeline above is ~17 launches and, behind the accumulate pass, two reduction passes with chains of 16 + ~8 group
// additions one after another: 0.65 ms for 2^8 points, 0.9 ms for 2^12, whatever the chip could do beside (a group addition is 10-14
// Fq products in sequence on a lane: 11-17 us).  With the table every (point, window) pair is a point T[w n + i] of its own with a
// signed digit d, and  sum d T = sum_t 2^t ( sum over the pairs whose |d| has bit t of +-T ):  one PLANE per digit bit, each a plain
// sum.  msm_small_planes_kernel: one wave per (slot, plane) -- a lane adds the pairs of its stride whose digit has the bit (mixed
// additions), the wave sums its lanes by a tree in registers; msm_small_reduce_kernel: one wave per plane sums the slots' partial
// sums the same way and converts.  Two launches, each a few mixed / full additions per lane and a 64-lane tree whose additions run on
// quads of lanes (msm_wave_tree_sum: 7 passes of ~3 k instructions); the host epilogue is the weighted sum of <= 20 plane sums (one
// doubling and one addition each).  Same group element, so the same affine commitment.  0.245 ms at 2^8, 0.47 ms at 2^12.
// A BATCH of such commits (the rounds of a small MultilinearKZG::open against their level tables) takes the same two launches: the
// slots are dealt to the problems in proportion to their pairs, a problem's planes are its own digit width.
constexpr int MSM_SMALL_PROBS = 16;
struct MsmSmallArgs {
    const uint32_t* table;       // problem j: [tab_off[j] + w * stride[j] + i]: 2^(first bit of window w) * point i, affine, 28-bit limbs
    const uint64_t* scalars;     // problem j: n[j] of them from sc_off[j] on, Montgomery
    const uint8_t* inf;          // flags of the points (indexed like the scalars) or nullptr
    uint32_t* partials;          // [plane * total_slots + slot], XYZZ
    uint64_t* terms;             // [j * planes + plane], XYZZ in the arkworks layout (4 x 48 B)
    uint32_t nprob, planes, total_slots;
    uint32_t n[MSM_SMALL_PROBS], stride[MSM_SMALL_PROBS], W[MSM_SMALL_PROBS], hi[MSM_SMALL_PROBS], n_hi[MSM_SMALL_PROBS];
    uint32_t tab_off[MSM_SMALL_PROBS], sc_off[MSM_SMALL_PROBS], slot_first[MSM_SMALL_PROBS + 1];
};
static __global__ __launch_bounds__(64) void msm_small_planes_kernel(MsmSmallArgs a) {
    const uint32_t slot = blockIdx.x, plane = blockIdx.y, lane = threadIdx.x;
    uint32_t j = 0;
    while (j + 1 < a.nprob && slot >= a.slot_first[j + 1]) ++j;
    if (plane >= a.hi[j]) return;                                   // |digit| <= 2^(hi - 1): bits 0 .. hi - 1
    const uint32_t n = a.n[j], pairs = n * a.W[j], s = slot - a.slot_first[j], lanes_total = (a.slot_first[j + 1] - a.slot_first[j]) * 64;
    const uint32_t hi = a.hi[j], n_hi = a.n_hi[j];
    G1XyzzU acc = G1XyzzU::identity();
    for (uint32_t p = s * 64 + lane; p < pairs; p += lanes_total) {
        const uint32_t w = p / n, i = p - w * n;
        if (a.inf && a.inf[a.sc_off[j] + i]) continue;
        DigitStream ds(load_fr(a.scalars, (size_t)a.sc_off[j] + i).from_mont());
        int32_t d = 0;
        for (uint32_t v = 0; v <= w; ++v) d = ds.next(v < n_hi ? hi : hi - 1);
        const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
        if ((mag >> plane) & 1u) g1u_madd(acc, load_affine_u(a.table, (size_t)a.tab_off[j] + (size_t)w * a.stride[j] + i), d < 0);
    }
    msm_wave_tree_sum(acc, 64);
    if (lane == 0) store_xyzz_u(a.partials, (size_t)plane * a.total_slots + slot, acc);
}
static __global__ __launch_bounds__(64) void msm_small_reduce_kernel(MsmSmallArgs a) {
    const uint32_t plane = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    if (plane >= a.hi[j]) return;
    const uint32_t s0 = a.slot_first[j], ns = a.slot_first[j + 1] - s0;
    G1XyzzU acc = G1XyzzU::identity();
    for (uint32_t s = lane; s < ns; s += 64) {
        const G1XyzzU v = load_xyzz_u(a.partials, (size_t)plane * a.total_slots + s0 + s);
        g1u_add(acc, v);
    }
    msm_wave_tree_sum(acc, 64);
    if (lane == 0) {
        uint64_t* o = a.terms + 24 * ((size_t)j * a.planes + plane);
        store_fq(o, fqu_to_ark(acc.x));
        store_fq(o + 6, fqu_to_ark(acc.y));
        store_fq(o + 12, fqu_to_ark(acc.zz));
        store_fq(o + 18, fqu_to_ark(acc.zzz));
    }
}

// ---- shifted-SRS table (zkhip_srs_precompute) -------------------------------------------------------------
static __global__ __launch_bounds__(MSM_BLOCK) void msm_clear_inf_kernel(const uint8_t* __restrict__ inf, size_t n, uint32_t* __restrict__ table) {
    const size_t i = (size_t)blockIdx.x * MSM_BLOCK + threadIdx.x;
    if (i >= n || !inf[i]) return;
    store_fqu(table + 32 * i, FqU::zero());
    store_fqu(table + 32 * i + 16, FqU::zero());
}
// out[i] = 2^c * in[i] as XYZZ in the arkworks limb layout (for the batched affine conversion): c doublings of a
// table entry (internal affine layout; all-zero coordinates = the identity of an SRS point at infinity).
static __global__ __launch_bounds__(MSM_BLOCK) void msm_shift_points_kernel(const uint32_t* __restrict__ in, size_t n, uint32_t c,
                                                                     uint64_t* __restrict__ out_xyzz) {
    const size_t i = (size_t)blockIdx.x * MSM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const G1AffineU p = load_affine_u(in, i);
    uint64_t* o = out_xyzz + 24 * i;
    if (p.x.all_zero() && p.y.all_zero()) {
        const Fq z = Fq::zero();
        store_fq(o, z); store_fq(o + 6, z); store_fq(o + 12, z); store_fq(o + 18, z);
        return;
    }
    G1XyzzU acc = g1u_double_affine(p);
    for (uint32_t k = 1; k < c; ++k) acc = g1u_double(acc);
    store_fq(o, fqu_to_ark(acc.x));
    store_fq(o + 6, fqu_to_ark(acc.y));
    store_fq(o + 12, fqu_to_ark(acc.zz));
    store_fq(o + 18, fqu_to_ark(acc.zzz));
}

// ---- level tables of a SMALL SRS in one go (zkhip_srs_level_tables, n <= 2^12) ----------------------------------------------------------
// The window-by-window build (shift, batched affine conversion, layout conversion: three launches per window, ~30 windows per level,
// a dozen levels) is latency from end to end: 0.3-0.4 s for a table of a few MiB.  Here a lane owns a point of a level and walks its
// windows itself (all the doublings of a point: ~256, ~2 ms), every window's XYZZ goes to ONE batched affine conversion.
struct MsmSmallTabArgs {
    uint32_t n_levels, total_points;
    uint32_t h[MSM_SMALL_PROBS], pt_off[MSM_SMALL_PROBS], tab_off[MSM_SMALL_PROBS], W[MSM_SMALL_PROBS], hi[MSM_SMALL_PROBS], n_hi[MSM_SMALL_PROBS];
};
static __global__ __launch_bounds__(64) void msm_small_level_windows_kernel(MsmSmallTabArgs a, const uint32_t* __restrict__ points_u /* the levels' points, internal affine layout, in level order */,
                                                                     uint64_t* __restrict__ out_xyzz /* [table entry] */) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t >= a.total_points) return;
    uint32_t j = 0;
    while (j + 1 < a.n_levels && t >= a.pt_off[j + 1]) ++j;
    const uint32_t i = t - a.pt_off[j], h = a.h[j];
    const G1AffineU p = load_affine_u(points_u, t);
    const bool zero = p.x.all_zero() && p.y.all_zero();
    G1XyzzU acc;
    acc.x = p.x; acc.y = p.y; acc.zz = zero ? FqU::zero() : FqU::one(); acc.zzz = acc.zz;
    for (uint32_t w = 0; w < a.W[j]; ++w) {
        if (w) {
            const uint32_t width = w - 1 < a.n_hi[j] ? a.hi[j] : a.hi[j] - 1;       // the window before this one
            if (w == 1 && !zero) { acc = g1u_double_affine(p); for (uint32_t k = 1; k < width; ++k) acc = g1u_double(acc); }
            else for (uint32_t k = 0; k < width; ++k) acc = g1u_double(acc);
        }
        uint64_t* o = out_xyzz + 24 * ((size_t)a.tab_off[j] + (size_t)w * h + i);
        store_fq(o, fqu_to_ark(acc.x));
        store_fq(o + 6, fqu_to_ark(acc.y));
        store_fq(o + 12, fqu_to_ark(acc.zz));
        store_fq(o + 18, fqu_to_ark(acc.zzz));
    }
}

}  // namespace zk

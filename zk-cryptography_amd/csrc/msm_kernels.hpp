// msm_kernels.hpp -- multi-scalar multiplication over BLS12-381 G1 for gfx950 (bucket method).
//
// Replaces the commit loops of the reference,
//     UnivariateKZG::commitment   kzg/src/univariate_kzg.rs:37-58
//     MultilinearKZG::commitment  kzg/src/multilinear_kzg.rs:33-48
// which compute sum_i srs[i].mul_bigint(coeff[i].into_bigint()) by double-and-add.  The group element is
// the same; the algorithm is Pippenger's:
//   1. digits:   scalars leave Montgomery form (into_bigint) and are recoded into signed base-2^c digits;
//                a histogram over (window, |digit|) buckets is built with global atomics;
//   2. scan:     exclusive prefix sum of the histogram -> bucket offsets;
//   3. scatter:  point indices (+ sign) are written bucket by bucket (counting sort);
//   4. accumulate: one lane per bucket adds its points (mixed XYZZ additions);
//   5. segments: every L consecutive buckets are folded by a local running sum into
//                S_s = sum B and A_s = sum (j+1) B  -- short dependency chains only;
//   6. terms:    per window, sum_s A_s and, for every bit k of the segment index, T_k = sum_{s: bit k} S_s
//                by workgroup tree reductions.  The window total is sum_s A_s + L * sum_k 2^k T_k.
//   7. the (#windows x #terms) points, each tagged with its power-of-two weight, go to the host, which runs
//      the final 255-step double-and-add chain (inherently serial; a few hundred group operations).
// MSM is integer-ALU bound (about 10 Fq products of ~900 instructions per added point), not HBM bound.
#pragma once
#include "g1.hpp"
#include "g1u.hpp"

namespace zk {

constexpr int MSM_BLOCK = 256;
constexpr int MSM_SEG_LOG = 3;              // L = 8 buckets per segment
constexpr int MSM_SEG = 1 << MSM_SEG_LOG;
constexpr int MSM_MAX_WINDOWS = 64;

struct MsmPlan {
    uint32_t c;          // window bits
    uint32_t n_windows;  // ceil(256 / c)
    uint32_t nb;         // buckets per window = 2^(c-1); bucket i holds digit magnitude i+1
    uint32_t ns;         // segments per window = nb / L
    uint32_t n_bits;     // bits of the segment index = c - 1 - log2 L
    uint32_t n_terms;    // 1 + n_bits
    uint32_t sub_bits;   // low bits of the bucket index resolved inside a partition (<= 8)
    uint32_t parts_pw;   // partitions per window = nb >> sub_bits
    uint32_t n_parts;    // n_windows * parts_pw (<= 2048)
};

// Signed base-2^c digit stream of a canonical scalar (8 x u32, little endian); digits lie in [-nb, nb].
// The scalar is consumed by shifting (no dynamically indexed registers).
struct DigitStream {
    Fr v;
    uint32_t carry;
    __device__ __forceinline__ explicit DigitStream(const Fr& canon) : v(canon), carry(0) {}
    __device__ __forceinline__ int32_t next(const MsmPlan& pl) {
        const uint32_t c = pl.c;
        uint32_t raw = (v.l[0] & ((1u << c) - 1)) + carry;
#pragma unroll
        for (int i = 0; i < 7; ++i) v.l[i] = (v.l[i] >> c) | (v.l[i + 1] << (32 - c));
        v.l[7] >>= c;
        if (raw > pl.nb) { carry = 1; return (int32_t)raw - (int32_t)(1u << c); }
        carry = 0;
        return (int32_t)raw;
    }
};

// pass 1: histogram
static __global__ __launch_bounds__(MSM_BLOCK) void msm_hist_kernel(const uint64_t* __restrict__ scalars,
                                                             const uint8_t* __restrict__ inf, size_t n, MsmPlan pl,
                                                             uint32_t* __restrict__ counts) {
    const size_t stride = (size_t)gridDim.x * MSM_BLOCK;
    for (size_t i = (size_t)blockIdx.x * MSM_BLOCK + threadIdx.x; i < n; i += stride) {
        if (inf && inf[i]) continue;
        DigitStream ds(load_fr(scalars, i).from_mont());
        for (uint32_t w = 0; w < pl.n_windows; ++w) {
            const int32_t d = ds.next(pl);
            if (d == 0) continue;
            const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
            atomicAdd(&counts[w * pl.nb + mag - 1], 1u);
        }
    }
}

// pass 2: exclusive scan of `total` counters in three small launches (tile sums, scan of the tile sums, tile scans)
constexpr int SCAN_TILE = 1024;   // counters per workgroup (4 per lane)
static __global__ __launch_bounds__(MSM_BLOCK) void msm_scan_tiles_kernel(const uint32_t* __restrict__ counts, uint32_t total,
                                                                          uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t red[MSM_BLOCK / 64];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (base + u < total) s += counts[base + u];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// exclusive scan of n_tiles (<= 64 K) tile sums by one workgroup, in place
static __global__ __launch_bounds__(1024) void msm_scan_top_kernel(uint32_t* __restrict__ tile_sums, uint32_t n_tiles) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n_tiles + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, n_tiles);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += tile_sums[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = lo; i < hi; ++i) { uint32_t c = tile_sums[i]; tile_sums[i] = run; run += c; }
}
static __global__ __launch_bounds__(MSM_BLOCK) void msm_scan_finish_kernel(const uint32_t* __restrict__ counts, uint32_t total,
                                                                           const uint32_t* __restrict__ tile_offsets,
                                                                           uint32_t* __restrict__ offsets,
                                                                           uint32_t* __restrict__ cursor) {
    __shared__ uint32_t wsum[MSM_BLOCK / 64];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    uint32_t c[4], s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { c[u] = (base + u < total) ? counts[base + u] : 0; s += c[u]; }
    // inclusive scan of the lanes' sums inside the wave, then across the 4 waves
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t v = __shfl_up(incl, d, 64); if ((int)(threadIdx.x & 63) >= d) incl += v; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) woff += wsum[w];
    uint32_t run = tile_offsets[blockIdx.x] + woff + incl - s;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (base + u < total) { offsets[base + u] = run; cursor[base + u] = run; }
        run += c[u];
    }
}

// Bucket processing order: buckets sorted by their point count, heaviest first, so that the 64 lanes of a wave
// walk lists of (nearly) equal length.  Counting sort over the clamped count.
constexpr uint32_t MSM_COUNT_BINS = 1024;
// (counts cluster around their mean, so the bins are few and hot: aggregate in LDS, one global atomic per bin and workgroup)
static __global__ __launch_bounds__(MSM_BLOCK) void msm_order_hist_kernel(const uint32_t* __restrict__ counts, uint32_t n_buckets,
                                                                          uint32_t* __restrict__ bins) {
    __shared__ uint32_t local[MSM_COUNT_BINS];
    for (uint32_t i = threadIdx.x; i < MSM_COUNT_BINS; i += MSM_BLOCK) local[i] = 0;
    __syncthreads();
    const uint32_t b = blockIdx.x * MSM_BLOCK + threadIdx.x;
    if (b < n_buckets) atomicAdd(&local[MSM_COUNT_BINS - 1 - min(counts[b], MSM_COUNT_BINS - 1)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < MSM_COUNT_BINS; i += MSM_BLOCK)
        if (local[i]) atomicAdd(&bins[i], local[i]);
}
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

// pass 3: counting-sort scatter; entry = point index | sign << 31
static __global__ __launch_bounds__(MSM_BLOCK) void msm_scatter_kernel(const uint64_t* __restrict__ scalars,
                                                                const uint8_t* __restrict__ inf, size_t n, MsmPlan pl,
                                                                uint32_t* __restrict__ cursor,
                                                                uint32_t* __restrict__ sorted) {
    const size_t stride = (size_t)gridDim.x * MSM_BLOCK;
    for (size_t i = (size_t)blockIdx.x * MSM_BLOCK + threadIdx.x; i < n; i += stride) {
        if (inf && inf[i]) continue;
        DigitStream ds(load_fr(scalars, i).from_mont());
        for (uint32_t w = 0; w < pl.n_windows; ++w) {
            const int32_t d = ds.next(pl);
            if (d == 0) continue;
            const bool neg = d < 0;
            const uint32_t mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            const uint32_t pos = atomicAdd(&cursor[w * pl.nb + mag - 1], 1u);
            sorted[pos] = (uint32_t)i | (neg ? 0x80000000u : 0u);
        }
    }
}

// ---- counting sort in two levels, without global atomics -------------------------------------------------
// Level 1 splits the (point, window) pairs by (window, high bits of the bucket) into <= 2048 partitions; a workgroup
// counts its tile in LDS (sort_count), a column scan turns the per-workgroup counts into exclusive write positions
// (sort_bases), and the same tile walk scatters 8-byte items (sort_scatter).  Level 2 runs one workgroup per
// partition: the low <= 8 bucket bits are resolved in LDS, which also yields every bucket's count and offset.
constexpr int SORT_TILE = 2048;          // scalars per workgroup tile (8 per lane)
constexpr int SORT_MAX_PARTS = 2048;

__device__ __forceinline__ uint32_t msm_partition_of(uint32_t w, uint32_t mag, const MsmPlan& pl) {
    return w * pl.parts_pw + ((mag - 1) >> pl.sub_bits);
}

static __global__ __launch_bounds__(MSM_BLOCK) void msm_sort_count_kernel(const uint64_t* __restrict__ scalars,
                                                                          const uint8_t* __restrict__ inf, size_t n, MsmPlan pl,
                                                                          uint32_t* __restrict__ wg_counts) {
    __shared__ uint32_t local[SORT_MAX_PARTS];
    for (uint32_t i = threadIdx.x; i < pl.n_parts; i += MSM_BLOCK) local[i] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    for (uint32_t u = 0; u < SORT_TILE / MSM_BLOCK; ++u) {
        const size_t i = base + u * MSM_BLOCK + threadIdx.x;
        if (i >= n || (inf && inf[i])) continue;
        DigitStream ds(load_fr(scalars, i).from_mont());
        for (uint32_t w = 0; w < pl.n_windows; ++w) {
            const int32_t d = ds.next(pl);
            if (d == 0) continue;
            atomicAdd(&local[msm_partition_of(w, d < 0 ? (uint32_t)(-d) : (uint32_t)d, pl)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < pl.n_parts; i += MSM_BLOCK) wg_counts[(size_t)blockIdx.x * pl.n_parts + i] = local[i];
}

// per partition: exclusive scan of the workgroups' counts (in place) and the partition total
static __global__ __launch_bounds__(MSM_BLOCK) void msm_sort_bases_kernel(uint32_t* __restrict__ wg_counts, uint32_t n_wgs,
                                                                          uint32_t n_parts, uint32_t* __restrict__ part_count) {
    const uint32_t p = blockIdx.x * MSM_BLOCK + threadIdx.x;
    if (p >= n_parts) return;
    uint32_t run = 0;
    for (uint32_t g = 0; g < n_wgs; ++g) {
        const uint32_t c = wg_counts[(size_t)g * n_parts + p];
        wg_counts[(size_t)g * n_parts + p] = run;
        run += c;
    }
    part_count[p] = run;
}
// exclusive scan of <= 2048 partition totals by one workgroup; part_off[n_parts] = grand total
static __global__ __launch_bounds__(1024) void msm_sort_part_scan_kernel(const uint32_t* __restrict__ part_count, uint32_t n_parts,
                                                                         uint32_t* __restrict__ part_off) {
    __shared__ uint32_t part[1024];
    const uint32_t a = 2 * threadIdx.x, b = a + 1;
    const uint32_t va = a < n_parts ? part_count[a] : 0, vb = b < n_parts ? part_count[b] : 0;
    part[threadIdx.x] = va + vb;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t excl = part[threadIdx.x] - (va + vb);
    if (a < n_parts) part_off[a] = excl;
    if (b < n_parts) part_off[b] = excl + va;
    if (threadIdx.x == 1023) part_off[n_parts] = part[1023];
}

static __global__ __launch_bounds__(MSM_BLOCK) void msm_sort_scatter_kernel(const uint64_t* __restrict__ scalars,
                                                                            const uint8_t* __restrict__ inf, size_t n, MsmPlan pl,
                                                                            const uint32_t* __restrict__ wg_bases,
                                                                            const uint32_t* __restrict__ part_off,
                                                                            uint2* __restrict__ items) {
    __shared__ uint32_t cur[SORT_MAX_PARTS];
    for (uint32_t i = threadIdx.x; i < pl.n_parts; i += MSM_BLOCK)
        cur[i] = part_off[i] + wg_bases[(size_t)blockIdx.x * pl.n_parts + i];
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    const uint32_t sub_mask = (1u << pl.sub_bits) - 1;
    for (uint32_t u = 0; u < SORT_TILE / MSM_BLOCK; ++u) {
        const size_t i = base + u * MSM_BLOCK + threadIdx.x;
        if (i >= n || (inf && inf[i])) continue;
        DigitStream ds(load_fr(scalars, i).from_mont());
        for (uint32_t w = 0; w < pl.n_windows; ++w) {
            const int32_t d = ds.next(pl);
            if (d == 0) continue;
            const bool neg = d < 0;
            const uint32_t mag = neg ? (uint32_t)(-d) : (uint32_t)d;
            const uint32_t pos = atomicAdd(&cur[msm_partition_of(w, mag, pl)], 1u);
            items[pos] = make_uint2((uint32_t)i | (neg ? 0x80000000u : 0u), (mag - 1) & sub_mask);
        }
    }
}

// level 2: one workgroup per partition; writes the final order plus counts / offsets of its 2^sub_bits buckets
static __global__ __launch_bounds__(MSM_BLOCK) void msm_sort_local_kernel(const uint2* __restrict__ items,
                                                                          const uint32_t* __restrict__ part_off, MsmPlan pl,
                                                                          uint32_t* __restrict__ sorted,
                                                                          uint32_t* __restrict__ counts,
                                                                          uint32_t* __restrict__ offsets) {
    __shared__ uint32_t bins[256];
    __shared__ uint32_t scan[256];
    const uint32_t p = blockIdx.x;
    const uint32_t lo = part_off[p], hi = part_off[p + 1];
    const uint32_t n_sub = 1u << pl.sub_bits;
    if (threadIdx.x < 256) bins[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t q = lo + threadIdx.x; q < hi; q += MSM_BLOCK) atomicAdd(&bins[items[q].y], 1u);
    __syncthreads();
    // exclusive scan of the (<= 256) bins
    const uint32_t v = threadIdx.x < n_sub ? bins[threadIdx.x] : 0;
    scan[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t t = threadIdx.x >= d ? scan[threadIdx.x - d] : 0;
        __syncthreads();
        scan[threadIdx.x] += t;
        __syncthreads();
    }
    const uint32_t excl = scan[threadIdx.x] - v;
    if (threadIdx.x < n_sub) {
        const uint32_t w = p / pl.parts_pw, top = p % pl.parts_pw;
        const uint32_t bucket = w * pl.nb + (top << pl.sub_bits) + threadIdx.x;
        counts[bucket] = v;
        offsets[bucket] = lo + excl;
        bins[threadIdx.x] = lo + excl;      // becomes the write cursor
    }
    __syncthreads();
    for (uint32_t q = lo + threadIdx.x; q < hi; q += MSM_BLOCK) {
        const uint2 it = items[q];
        sorted[atomicAdd(&bins[it.y], 1u)] = it.x;
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

// pass 4: one lane per bucket (in the order given by `order`: heaviest buckets first, equal lengths inside a wave)
static __global__ __launch_bounds__(MSM_BLOCK) void msm_accumulate_kernel(const uint32_t* __restrict__ points,
                                                                   const uint32_t* __restrict__ sorted,
                                                                   const uint32_t* __restrict__ offsets,
                                                                   const uint32_t* __restrict__ counts,
                                                                   const uint32_t* __restrict__ order,
                                                                   uint32_t n_buckets, uint32_t* __restrict__ buckets) {
    const uint32_t t = blockIdx.x * MSM_BLOCK + threadIdx.x;
    if (t >= n_buckets) return;
    const uint32_t b = order[t];
    const uint32_t start = offsets[b], cnt = counts[b];
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
    store_xyzz_u(buckets, b, acc);
}

// pass 5: one lane per segment of L buckets: S = sum_j B_j, A = sum_j (j+1) B_j  (running sum from the top)
static __global__ __launch_bounds__(MSM_BLOCK) void msm_segment_kernel(const uint32_t* __restrict__ buckets,
                                                                uint32_t n_segments, uint32_t* __restrict__ seg_s,
                                                                uint32_t* __restrict__ seg_a) {
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

// pass 6: one workgroup per (window, term).  term 0: sum of A_s over the window's segments;
// term 1+k: sum of S_s over the segments whose index has bit k set.  The result leaves in the arkworks layout
// (XYZZ, 4 x 48 B) for the host epilogue.
static __global__ __launch_bounds__(MSM_BLOCK) void msm_terms_kernel(const uint32_t* __restrict__ seg_s,
                                                              const uint32_t* __restrict__ seg_a, MsmPlan pl,
                                                              uint64_t* __restrict__ terms) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(zk_dyn_lds);   // MSM_BLOCK x 64 u32
    const uint32_t w = blockIdx.x / pl.n_terms, t = blockIdx.x % pl.n_terms;
    const uint32_t* src = (t == 0 ? seg_a : seg_s) + (size_t)w * pl.ns * 64;
    G1XyzzU acc = G1XyzzU::identity();
    for (uint32_t s = threadIdx.x; s < pl.ns; s += MSM_BLOCK) {
        if (t != 0 && !((s >> (t - 1)) & 1)) continue;
        G1XyzzU v = load_xyzz_u(src, s);
        g1u_add(acc, v);
    }
    store_xyzz_u(lds, threadIdx.x, acc);
    __syncthreads();
    for (int d = MSM_BLOCK / 2; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            G1XyzzU o = load_xyzz_u(lds, threadIdx.x + d);
            g1u_add(acc, o);
            store_xyzz_u(lds, threadIdx.x, acc);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint64_t* o = terms + 24 * (size_t)blockIdx.x;
        store_fq(o, fqu_to_ark(acc.x));
        store_fq(o + 6, fqu_to_ark(acc.y));
        store_fq(o + 12, fqu_to_ark(acc.zz));
        store_fq(o + 18, fqu_to_ark(acc.zzz));
    }
}

}  // namespace zk

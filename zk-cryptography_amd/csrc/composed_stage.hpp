// composed_stage.hpp -- TWO rounds of a composed sumcheck per pass over the tables, for claims whose terms are products of two
// tables (+ an optional additive table): ComposedSumcheck::prove with K = 2 (sumcheck/src/composed/composed_sumcheck.rs:32-67,
// the reference's bench shape) and the GKR layer claims of MultiComposedSumcheckProver::prove_partial
// (sumcheck/src/composed/multi_composed_sumcheck.rs:64-121).
//
// A product of tables does not commute with block sums, but it is BILINEAR in them.  Cut both tables of a term into the four blocks
// of the next two variables (index = (a, j), a = the two most significant bits): with the cross-block sums
//     C[a][b] = sum_j A[a][j] * B[b][j]   (16 per term)        L[a] = sum_j lin[a][j]   (4, additive table)
// the next two round polynomials are functions of C and L alone -- with C_ab[x] = C[2 a + x][2 b + x], l_0(t) = 1 - t, l_1(t) = t:
//     round 1:  p(t) = sum_x sum_{a,b} l_a(t) l_b(t) C_ab[x]           + sum_x sum_a l_a(t) L[2 a + x]
//     bind r1:  C'[x][y] = sum_{a,b} l_a(r1) l_b(r1) C[2 a + x][2 b + y],     L'[x] = sum_a l_a(r1) L[2 a + x]
//     round 2:  p(t) = sum_{x,y} l_x(t) l_y(t) C'[x][y]                + sum_x l_x(t) L'[x]
// and the tables are then folded by both challenges at once, T'[j] = sum_a l_{a1}(r1) l_{a0}(r2) T[a][j].  Field arithmetic is
// exact, so the evaluations p(0), p(1), p(2) of every term -- what close_round interpolates, drops zeros from and absorbs -- are the
// round-by-round ones bit for bit.  Per two rounds: one pass that reads the tables (cross sums), ONE serial kernel (two transcript
// rounds), one pass that reads them again and writes a quarter -- instead of two passes that each read and write plus two serial
// kernels; a sharded proof exchanges one record per two rounds.
#pragma once
#include "composed_kernels.hpp"

namespace zk {

constexpr int CST_VALS = 20;          // per term: C[4][4] (index 4 a + b), then L[4]
constexpr int CST_BLOCK = 256;
constexpr int CST_MAX_GRID = 256;     // records per stage: one workgroup per CU
constexpr int CST_CROSS_BLOCK = 1024;  // 16 waves: four per SIMD (a lone wave issues a v_mad_u64_u32 every 10.7 cycles, two or more every 5)

// Cross-block sums of every term (blockIdx.y): lane (pq = lane % 16 -> a = pq / 4, b = pq % 4) multiplies A[a][j] * B[b][j] for the
// j of its slot (64 slots per workgroup) into an unreduced 17-limb accumulator -- 64 mads + carry adds per product, one 9-word
// reduction per lane at the end; each entry is loaded by the four lanes that use it (the L1 serves three of them).
// partials[(blockIdx.x * n_terms + term) * CST_VALS + v].
static __global__ __launch_bounds__(CST_CROSS_BLOCK) void composed_cross2_kernel(MultiTablePtrs mp, size_t n, uint32_t n_terms,
                                                                           uint64_t* __restrict__ partials) {
    __shared__ Fr red[CST_CROSS_BLOCK / 64][20];
    const uint32_t term = blockIdx.y;
    const TablePtrs& tp = mp.t[term];
    const size_t m = n / 4;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t pq = lane & 15, a = pq >> 2, b = pq & 3;
    const uint32_t slot = wave * 4 + (lane >> 4);
    const size_t per = (m + gridDim.x - 1) / gridDim.x;
    const size_t j0 = (size_t)blockIdx.x * per, j1 = j0 + per < m ? j0 + per : m;
    const uint64_t* pa = tp.in[0] + 4 * ((size_t)a * m);
    const uint64_t* pb = tp.in[1] + 4 * ((size_t)b * m);
    const uint64_t* pl = (tp.lin_in && b == 0) ? tp.lin_in + 4 * ((size_t)a * m) : nullptr;
    WideAcc acc;
    acc.clear();
    Fr ls = Fr::zero();
    constexpr int U = 2;
    constexpr uint32_t SLOTS = CST_CROSS_BLOCK / 16;
    for (size_t j = j0 + slot; j < j1; j += SLOTS * U) {
        Fr x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t ju = j + SLOTS * (size_t)u;
            const bool ok = ju < j1;
            x[u] = ok ? load_fr(pa, ju) : Fr::zero();
            y[u] = ok ? load_fr(pb, ju) : Fr::zero();
            z[u] = (ok && pl) ? load_fr(pl, ju) : Fr::zero();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.mac(x[u], y[u]); ls = ls + z[u]; }
    }
    // x * 2^-288 of two Montgomery residues is the product's residue times 2^-32: one product with mont(2^32) restores it
    Fr c = wide_reduce(acc.lo, acc.hi) * fr_mont_2_32();
    c = c + shfl_down_fr(c, 32);
    c = c + shfl_down_fr(c, 16);
    ls = ls + shfl_down_fr(ls, 32);
    ls = ls + shfl_down_fr(ls, 16);
    if (lane < 16) {
        red[wave][pq] = c;
        if (b == 0) red[wave][16 + a] = ls;
    }
    __syncthreads();
    if (threadIdx.x < CST_VALS) {
        Fr s = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < CST_CROSS_BLOCK / 64; ++w) s = s + red[w][threadIdx.x];
        store_fr(partials, ((size_t)blockIdx.x * n_terms + term) * CST_VALS + threadIdx.x, s);
    }
}

struct StageArgs {
    CloseArgs ca;                 // ca.round = the first of the two rounds
    uint32_t n_records;
    uint64_t* weights_out;        // 4 fold weights l_{a1}(r1) l_{a0}(r2) * 2^32 (Montgomery): the unreduced fold's operand
};
// Sums the workgroup (or rank) records and runs the two rounds.
static __global__ __launch_bounds__(CST_BLOCK) void composed_stage_close_kernel(const uint64_t* __restrict__ partials, StageArgs sa) {
    __shared__ CloseShared sh;
    __shared__ Sha256State trs;
    __shared__ Fr vals[CMP_MAX_TERMS][CST_VALS];
    __shared__ Fr bound[CMP_MAX_TERMS][6];       // C'[x][y] at 2 x + y, then L'[0], L'[1]
    __shared__ Fr r1m;
    const CloseArgs& ca = sa.ca;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t P = ca.meta.n_terms, n_vals = P * CST_VALS;
    if (ca.first != 1 && tid < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&trs)[tid] = reinterpret_cast<const uint32_t*>(&ca.st->transcript)[tid];
    close_preload(sh, ca.meta, ca.st);
    for (uint32_t v = wave; v < n_vals; v += CST_BLOCK / 64) {
        Fr s = Fr::zero();
        for (uint32_t rdx = lane; rdx < sa.n_records; rdx += 256) {
            Fr x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = rdx + 64 * u < sa.n_records ? load_fr(partials, (size_t)(rdx + 64 * u) * n_vals + v) : Fr::zero();
            s = s + ((x[0] + x[1]) + (x[2] + x[3]));
        }
        s = wave_reduce_fr(s);
        if (lane == 0) vals[v / CST_VALS][v % CST_VALS] = s;
    }
    __syncthreads();
    // ---- round 1: p(0), p(1), p(2) of every term from C and L
    if (tid < 3 * P) {
        const uint32_t p = tid / 3, e = tid % 3;
        const Fr* C = vals[p];
        const Fr* L = vals[p] + 16;
        Fr v;
        if (e == 0) v = (C[0] + C[5]) + (L[0] + L[1]);                        // C[0][0] + C[1][1]
        else if (e == 1) v = (C[10] + C[15]) + (L[2] + L[3]);                 // C[2][2] + C[3][3]
        else {
            // sum_x C[x][x] - 2 C[x][2+x] - 2 C[2+x][x] + 4 C[2+x][2+x]  -  L[x] + 2 L[2+x]
            const Fr ll = C[0] + C[5], hh = C[10] + C[15], lh = (C[2] + C[7]) + (C[8] + C[13]);
            const Fr hh2 = hh + hh, lh2 = lh + lh;
            v = (ll + (hh2 + hh2)) - lh2;
            const Fr lhi = L[2] + L[3];
            v = v + ((lhi + lhi) - (L[0] + L[1]));
        }
        sh.evals[ca.meta.rec_off[p] + e] = v;
    }
    __syncthreads();
    close_round(sh, ca.meta, ca.st, ca.sum, &trs, ca.round, ca.first, ca.round_out, ca.challenges);
    // ---- bind the first variable.  A lone wave pays ~0.9 us per product whatever the dependencies, so the bind is laid out THREE products
    // deep instead of seven: every lane converts the challenge for itself, lane (p, i, a, b) multiplies l_a(r1) l_b(r1) and then its one
    // entry of C (or L), and the four (two) partial products of a bound value are added from LDS.
    __shared__ Fr bprod[CMP_MAX_TERMS][24];     // per term: 16 products for C'[x][y] (4 each), 4 for L'[x] (2 each, padded to 4)
    const Fr r1 = fr_to_mont_outlined(sh.challenge_canon);
    if (tid == 0) r1m = r1;
    if (tid < 24 * P) {
        const uint32_t p = tid / 24, q = tid % 24;
        const Fr* C = vals[p];
        const Fr* L = vals[p] + 16;
        const Fr l0 = Fr::one() - r1;
        Fr v = Fr::zero();
        if (q < 16) {
            const uint32_t i = q >> 2, ab = q & 3, x = i >> 1, y = i & 1, a = ab >> 1, b = ab & 1;
            const Fr w = (a ? r1 : l0) * (b ? r1 : l0);
            v = C[4 * (2 * a + x) + 2 * b + y] * w;
        } else if (q < 20) {
            const uint32_t x = (q - 16) >> 1, a = (q - 16) & 1;
            v = L[2 * a + x] * (a ? r1 : l0);
        }
        bprod[p][q] = v;
    }
    __syncthreads();
    if (tid < 6 * P) {
        const uint32_t p = tid / 6, i = tid % 6;
        const Fr* B = bprod[p];
        bound[p][i] = i < 4 ? (B[4 * i] + B[4 * i + 1]) + (B[4 * i + 2] + B[4 * i + 3]) : B[16 + 2 * (i - 4)] + B[16 + 2 * (i - 4) + 1];
    }
    __syncthreads();
    // ---- round 2 on the bound sums
    if (tid < 3 * P) {
        const uint32_t p = tid / 3, e = tid % 3;
        const Fr* B = bound[p];
        Fr v;
        if (e == 0) v = B[0] + B[4];
        else if (e == 1) v = B[3] + B[5];
        else {
            const Fr lh = B[1] + B[2], hh2 = B[3] + B[3];
            v = (B[0] + (hh2 + hh2)) - (lh + lh);
            v = v + ((B[5] + B[5]) - B[4]);
        }
        sh.evals[ca.meta.rec_off[p] + e] = v;
    }
    __syncthreads();
    close_round(sh, ca.meta, ca.st, ca.sum, &trs, ca.round + 1, 0u, ca.round_out, ca.challenges);
    if (tid < 4) {
        const Fr r2 = fr_to_mont_outlined(sh.challenge_canon), r1v = r1m;
        const Fr one = Fr::one();
        const Fr f1 = (tid >> 1) ? r1v : one - r1v, f2 = (tid & 1) ? r2 : one - r2;
        store_fr(sa.weights_out, tid, (f1 * f2) * fr_mont_2_32());
    }
    if (tid < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&ca.st->transcript)[tid] = reinterpret_cast<const uint32_t*>(&trs)[tid];
}

// the records of a stage summed into one (what a rank contributes to the exchange of the sharded protocol)
static __global__ __launch_bounds__(CST_BLOCK) void composed_stage_reduce_kernel(const uint64_t* __restrict__ partials, uint32_t n_records,
                                                                                 uint32_t n_vals, uint64_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t v = wave; v < n_vals; v += CST_BLOCK / 64) {
        Fr s = Fr::zero();
        for (uint32_t rdx = lane; rdx < n_records; rdx += 64) s = s + load_fr(partials, (size_t)rdx * n_vals + v);
        s = wave_reduce_fr(s);
        if (lane == 0) store_fr(out, v, s);
    }
}

// T'[j] = sum_{a < 4} w[a] * T[a * m + j] for every table (blockIdx.y) of the claim: the fold by both challenges of the stage
struct Fold2Tables {
    const uint64_t* in[CMP_MAX_TERMS * 3];
    uint64_t* out[CMP_MAX_TERMS * 3];
};
static __global__ __launch_bounds__(CST_BLOCK) void composed_fold2_kernel(Fold2Tables ft, size_t m, const uint64_t* __restrict__ weights) {
    const uint64_t* in = ft.in[blockIdx.y];
    uint64_t* out = ft.out[blockIdx.y];
    Fr w[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) w[a] = load_fr(weights, a);
    const size_t stride = (size_t)gridDim.x * CST_BLOCK;
    for (size_t j = (size_t)blockIdx.x * CST_BLOCK + threadIdx.x; j < m; j += stride) {
        Fr t[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) t[a] = load_fr(in, (size_t)a * m + j);
        WideAcc acc;
        acc.clear();
#pragma unroll
        for (int a = 0; a < 4; ++a) acc.mac(w[a], t[a]);
        store_fr(out, j, wide_reduce(acc.lo, acc.hi));
    }
}

}  // namespace zk

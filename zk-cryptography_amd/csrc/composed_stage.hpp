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
#include "mfma_fold.hpp"

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

// ---- the cross-block sums on the matrix cores ---------------------------------------------------------------------------------
// C[a][b] = sum_j A[a][j] * B[b][j] is data x data -- no fixed operand to lay out as a Toeplitz matrix (mfma_fold.hpp) -- but byte by
// byte it is an outer product accumulated over j:  G[d][i] = sum_j A[a][j].byte[d] * B[b][j].byte[i]  (32 x 32), a GEMM whose
// contraction index is j, and C[a][b] = sum_c 2^(8c) sum_{d+i=c} G[d][i].  The MFMA wants, per lane, 16 consecutive K of one row:
// byte d of 16 consecutive entries, i.e. the table TRANSPOSED -- which gfx950 does on the way out of LDS: ds_read_b64_tr_b8 reads, per
// 16 lanes, 8 rows of 16 bytes (row p at the addresses of lanes 2p and 2p + 1) and hands lane c column c (tools/probe_tr_b8.hip).  A
// tile of 32 entries staged in LDS as it lies in memory therefore yields the operand with two such reads per lane.
// One workgroup = 4 waves; wave a multiplies block a of the first table with the four blocks of the second: per step of 32 indices
// 8 KiB staged, 10 transposed reads and 4 MFMAs per wave.  Bytes are fed as u ^ 0x80 = u - 128 (int8 is signed); the true unsigned sums
// follow from the per-position byte sums (v_sad_u8):  sum ua ub = G + 128 UA[d] + 128 UB[i] - 16384 J.  At most 65536 indices per
// workgroup: |G| <= 2^30 and the true sums fit 32 bits.
constexpr int CSM_STEP = 32;           // indices per step = K of one MFMA
constexpr int CSM_TILE = 1024;         // bytes of one staged tile (32 entries)

// the five operands of a wave's step in one go -- block a of the first table, the four blocks of the second -- so that the ten
// transposed reads are in flight together (one wait instead of five LDS round trips per step)
typedef uint32_t csm_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void csm_operands(const unsigned char* buf, uint32_t wave, uint32_t lane, mf_v4i& xa, mf_v4i (&yb)[4]) {
    // lane (g = lane / 16, i = lane % 16): byte 16 (g % 2) + i of entries 16 (g / 2) .. + 15 of a tile
    const uint32_t g = lane >> 4, i = lane & 15;
    const uint32_t in_tile = 32 * (16 * (g >> 1) + (i >> 1)) + 16 * (g & 1) + 8 * (i & 1);
    const uint32_t a0 = (uint32_t)(uintptr_t)(buf + wave * CSM_TILE + in_tile);
    const uint32_t b0 = (uint32_t)(uintptr_t)(buf + 4 * CSM_TILE + in_tile);
    csm_u32x2 r[10];
    asm volatile(
        "ds_read_b64_tr_b8 %0, %10\n\tds_read_b64_tr_b8 %1, %10 offset:256\n\t"
        "ds_read_b64_tr_b8 %2, %11\n\tds_read_b64_tr_b8 %3, %11 offset:256\n\t"
        "ds_read_b64_tr_b8 %4, %11 offset:1024\n\tds_read_b64_tr_b8 %5, %11 offset:1280\n\t"
        "ds_read_b64_tr_b8 %6, %11 offset:2048\n\tds_read_b64_tr_b8 %7, %11 offset:2304\n\t"
        "ds_read_b64_tr_b8 %8, %11 offset:3072\n\tds_read_b64_tr_b8 %9, %11 offset:3328\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9])
        : "v"(a0), "v"(b0)
        : "memory");
    xa.x = (int)r[0].x; xa.y = (int)r[0].y; xa.z = (int)r[1].x; xa.w = (int)r[1].y;
#pragma unroll
    for (int b = 0; b < 4; ++b) { yb[b].x = (int)r[2 + 2 * b].x; yb[b].y = (int)r[2 + 2 * b].y; yb[b].z = (int)r[3 + 2 * b].x; yb[b].w = (int)r[3 + 2 * b].y; }
}
__device__ __forceinline__ uint32_t csm_bytesum(const mf_v4i& o, uint32_t acc) {
    acc = __builtin_amdgcn_sad_u8((uint32_t)o.x, 0u, acc);
    acc = __builtin_amdgcn_sad_u8((uint32_t)o.y, 0u, acc);
    acc = __builtin_amdgcn_sad_u8((uint32_t)o.z, 0u, acc);
    return __builtin_amdgcn_sad_u8((uint32_t)o.w, 0u, acc);
}
static __global__ __launch_bounds__(256) void composed_cross2_mfma_kernel(MultiTablePtrs mp, size_t n, uint32_t n_terms,
                                                                          uint64_t* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) unsigned char tiles[2][8 * CSM_TILE];      // double-buffered: A0..A3, B0..B3
    __shared__ uint32_t tmat[4][32 * 33];         // per wave: the true byte-product sums of one (a, b), [d][i] padded
    __shared__ uint32_t ua_sh[4][32], ub_sh[4][4][32];
    __shared__ unsigned long long cols[4][4][64]; // per wave and b: the 63 anti-diagonal sums
    __shared__ Fr red[4][4];
    __shared__ Fr lred[4][4];
    const uint32_t term = blockIdx.y;
    const TablePtrs& tp = mp.t[term];
    const size_t m = n / 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t per = ((m + gridDim.x - 1) / gridDim.x + CSM_STEP - 1) / CSM_STEP * CSM_STEP;     // a multiple of the step
    const size_t j0 = (size_t)blockIdx.x * per, j1 = j0 + per < m ? j0 + per : m;                  // m is a multiple of 32 (host)
    const uint32_t n_steps = j0 < j1 ? (uint32_t)((j1 - j0) / CSM_STEP) : 0;
    // this thread's two 16-byte chunks of a step's 8 KiB: chunk q = tid + 256 u -> tile q / 64, bytes 16 (q % 64) of it
    const unsigned char* src[2];
    uint32_t dst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t q = tid + 256 * u, tile = q >> 6, off = 16 * (q & 63);
        const uint64_t* base = tile < 4 ? tp.in[0] + 4 * ((size_t)tile * m) : tp.in[1] + 4 * ((size_t)(tile - 4) * m);
        src[u] = reinterpret_cast<const unsigned char*>(base) + 32 * j0 + off;
        dst[u] = tile * CSM_TILE + off;
    }
    mf_v16i acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = mf_v16i{0};
    uint32_t ua = 0, ub[4] = {0, 0, 0, 0};
    // loads run CSM_AHEAD steps ahead of the MFMAs (one step ahead left every step waiting for memory: 16 steps of a 2^20 claim took 16 us)
    constexpr int CSM_AHEAD = 4;
    mf_v4i ld[CSM_AHEAD][2];
#pragma unroll
    for (int q = 0; q < CSM_AHEAD; ++q)
        if ((uint32_t)q < n_steps) {
#pragma unroll
            for (int u = 0; u < 2; ++u) ld[q][u] = mfm_load_nt(src[u] + (size_t)q * (32 * CSM_STEP));
        }
    for (uint32_t s0 = 0; s0 < n_steps; s0 += CSM_AHEAD) {
#pragma unroll
      for (int q = 0; q < CSM_AHEAD; ++q) {
        const uint32_t s = s0 + q;
        if (s >= n_steps) break;
        unsigned char* buf = tiles[s & 1];
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<mf_v4i*>(buf + dst[u]) = ld[q][u];
        if (s + CSM_AHEAD < n_steps) {
#pragma unroll
            for (int u = 0; u < 2; ++u) ld[q][u] = mfm_load_nt(src[u] + (size_t)(s + CSM_AHEAD) * (32 * CSM_STEP));
        }
        __syncthreads();                     // the step's tiles are staged (the other buffer is free again: its readers passed this barrier)
        mf_v4i xa, yb[4];
        csm_operands(buf, wave, lane, xa, yb);
        ua = csm_bytesum(xa, ua);
        const mf_v4i xs = mfm_signed(xa);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            ub[b] = csm_bytesum(yb[b], ub[b]);
            acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xs, mfm_signed(yb[b]), acc[b], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    // ---- epilogue: per wave, the four (a = wave, b) sums as field elements
    const uint32_t col_i = lane & 31, hh = lane >> 5;
    const uint32_t jcount = (uint32_t)(j0 < j1 ? j1 - j0 : 0);
    {   // byte sums of both lane halves, by byte position
        const uint32_t ua_t = ua + (uint32_t)__shfl_xor((int)ua, 32, 64);
        if (lane < 32) ua_sh[wave][lane] = ua_t;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t t = ub[b] + (uint32_t)__shfl_xor((int)ub[b], 32, 64);
            if (lane < 32) ub_sh[wave][b][lane] = t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        // result register r = 4 q + t of lane (i, h): row d = 8 q + 4 h + t, column i
        const uint32_t ubi = ub_sh[wave][b][col_i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t d = 8 * (r >> 2) + 4 * hh + (r & 3);
            tmat[wave][d * 33 + col_i] = (uint32_t)acc[b][r] + 128u * (ua_sh[wave][d] + ubi) - 16384u * jcount;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane < 63) {                     // anti-diagonal c = lane
            unsigned long long sacc = 0;
#pragma unroll
            for (int d = 0; d < 32; ++d) {       // (all 32 rows, predicated: the loads are independent of each other and of the running sum)
                const int i = (int)lane - d;
                const uint32_t v = tmat[wave][d * 33 + (i & 31)];
                sacc += (i >= 0 && i < 32) ? v : 0u;
            }
            cols[wave][b][lane] = sacc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < 4) {                          // lane b: columns -> 17 limbs -> REDC -> back to the product's residue
        uint32_t x[18];
        unsigned long long carry = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            unsigned long long v = carry;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = 4 * g + t;
                if (c < 63) v += cols[wave][lane][c] << (8 * t);     // cols < 2^37: no overflow
            }
            x[g] = (uint32_t)v;
            carry = v >> 32;
        }
        x[16] = (uint32_t)carry;
        x[17] = 0;
        red[wave][lane] = wide_redc(x) * fr_mont_2_32();
    }
    // the additive table's block sums (plain additions)
    if (tp.lin_in) {
        Fr ls = Fr::zero();
        const uint64_t* pl = tp.lin_in + 4 * ((size_t)wave * m);       // wave a sums block a
        for (size_t j = j0 + lane; j < j1; j += 64) ls = ls + load_fr(pl, j);
        ls = wave_reduce_fr(ls);
        if (lane == 0) lred[wave][0] = ls;
    }
    __syncthreads();
    if (tid < 16) store_fr(partials, ((size_t)blockIdx.x * n_terms + term) * CST_VALS + tid, red[tid >> 2][tid & 3]);
    else if (tid < 20) store_fr(partials, ((size_t)blockIdx.x * n_terms + term) * CST_VALS + tid, tp.lin_in ? lred[tid - 16][0] : Fr::zero());
}

struct StageArgs {
    CloseArgs ca;                 // ca.round = the first of the two rounds
    uint32_t n_records;
    uint64_t* weights_out;        // 4 fold weights l_{a1}(r1) l_{a0}(r2) * 2^32 (Montgomery): the unreduced fold's operand
};
// Sums the workgroup (or rank) records and runs the two rounds.
static __global__ __launch_bounds__(CST_BLOCK) void composed_stage_close_kernel(const uint64_t* __restrict__ partials, StageArgs sa) {
    if (sa.ca.outer.dev && blockIdx.x == gridDim.x - 1) { outer_absorb_rounds(sa.ca.outer, sa.ca.round, 2); return; }   // the hasher workgroup
    __shared__ CloseShared sh;
    __shared__ Sha256State trs;
    __shared__ Fr vals[CMP_MAX_TERMS][CST_VALS];
    __shared__ Fr bound[CMP_MAX_TERMS][6];       // C'[x][y] at 2 x + y, then L'[0], L'[1]
    __shared__ Fr r1m;
    const CloseArgs& ca = sa.ca;
    const uint32_t tid = threadIdx.x;
    const uint32_t P = ca.meta.n_terms, n_vals = P * CST_VALS;
    if (ca.first != 1 && tid < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&trs)[tid] = reinterpret_cast<const uint32_t*>(&ca.st->transcript)[tid];
    close_preload(sh, ca.meta, ca.st);
    {
        // thread (value v = tid % n_vals, chunk = tid / n_vals) sums the records chunk, chunk + n_chunks, ...: neighbouring threads read
        // neighbouring values of one record (coalesced), eight loads in flight per thread; the chunks are added in LDS
        __shared__ Fr rsum[CST_BLOCK];
        const uint32_t n_chunks = CST_BLOCK / n_vals, v = tid % n_vals, chunk = tid / n_vals;
        Fr acc = Fr::zero();
        if (chunk < n_chunks) {
            for (uint32_t r0 = chunk; r0 < sa.n_records; r0 += 8 * n_chunks) {
                Fr x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = r0 + u * n_chunks < sa.n_records ? load_fr(partials, (size_t)(r0 + u * n_chunks) * n_vals + v) : Fr::zero();
                acc = acc + (((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7])));
            }
        }
        rsum[tid] = acc;
        __syncthreads();
        if (tid < n_vals) {
            Fr t = rsum[tid];
            for (uint32_t q = 1; q < n_chunks; ++q) t = t + rsum[q * n_vals + tid];
            vals[tid / CST_VALS][tid % CST_VALS] = t;
        }
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
    close_round(sh, ca, &trs, ca.round, ca.first);
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
    close_round(sh, ca, &trs, ca.round + 1, 0u);
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

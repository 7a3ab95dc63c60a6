// composed_dot.hpp -- the round kernel of a product term of FIVE tables with its LAST factor on the matrix cores.
// gfx950 only.
//
// composed_round_kernel<5> (composed_kernels.hpp) is bound by the issue of multiply-adds: per output pair 10 products fold the
// tables, 3 per pair of tables give that pair's quadratic, 6 the product Q_t of the two quadratics at t = 0..5, and the round polynomial's values
//     S(t) = sum_j Q_t[j] * (l[j] + t d[j]),   t = 0..5      ((l, d) = the last table)
// cost one more product per point and index: 6 of 28 products (6 of 18 in the first round).  That last step is a DOT PRODUCT over j --
// sum_j Q_t[j] l[j] + t sum_j Q_t[j] d[j] -- and byte by byte a dot product of field elements is a GEMM whose contraction index is j
// (composed_stage.hpp, the cross sums of the K = 2 stage form):
//     G[a][b] = sum_j Q_t[j].byte[a] * l[j].byte[b],      sum_j Q_t[j] l[j] = sum_c 2^(8c) sum_{a+b=c} G[a][b],
// one Montgomery reduction per workgroup, point and operand instead of one per index.  Here the operands are not tables in memory but
// values the lanes have just computed, so every lane writes its values Q_t and (l, d) into LDS as 32-byte entries -- the layout a table
// has in memory -- and the waves read them back transposed (ds_read_b64_tr_b8), exactly as composed_cross2_mfma_kernel does with staged
// tiles.  A workgroup of four waves stages 256 indices at a time (7 x 8 KiB).  Q_t has degree 4 in t: FIVE planes (t < 5) are staged and the
// sums at t = 5 follow by finite differences, one more product less per index.  The 10 accumulator tiles (plane x {l, d}) are split over
// the waves (2-3 tiles = 16-24 MFMAs of 32 indices per wave and 256 indices), so a wave holds 48 accumulator registers instead of the
// six running sums (48 registers) of the vector form: 238 registers, two waves per SIMD as before.
// Bytes are fed as u ^ 0x80 = u - 128 (int8 is signed); the true sums follow from the per-position byte sums (v_sad_u8):
// sum ua ub = G + 128 UA[a] + 128 UB[b] - 16384 J.  At most 65536 staged indices per workgroup (|G| <= 2^30; the host falls back to the
// vector form beyond).  EXACT: sum_j mont(Q_t[j], l[j]) and (sum_j Q_t[j] l[j]) 2^-256 are the same residue, the records hold canonical values.
// Where the time goes (2^22, first round, 512 workgroups): 397 us vector form (18 products per index) -> 293 us (11 products); without the
// MFMA phase and its two barriers per step 262 us -- the products themselves run ~8 % slower per product than in the vector form
// (all four waves of a workgroup load, compute and wait in step).  A term of THREE tables was built the same way (7 -> 3 products in the
// first round) and dropped: its rounds at the bench's 2^20 are 35-55 us long, less than the staging and the closing reduction cost.
#pragma once
#include "composed_stage.hpp"

namespace zk {

constexpr int CDT_ROWS = 256;                    // indices staged per step of a workgroup = its lanes
constexpr int CDT_PLANE = CDT_ROWS * 32;         // bytes of one staged operand (8 KiB)
// Q_t has degree K - 1 in t, so K planes (t = 0 .. K - 1) fix both sums as polynomials in t; their values at t = K follow by finite differences
__host__ __device__ constexpr size_t cdt_lds_bytes(int k) { return (size_t)(k + 2) * CDT_PLANE; }      // 56 KiB at K = 5
constexpr size_t CDT_MAX_PER_WG = 65536;         // staged indices per workgroup (int32 accumulators, 32-bit true sums)

__device__ __forceinline__ void cdt_store_row(unsigned char* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// The entries of table 0 (and, in the first round, table 1) behind output pair j, loaded ONE STEP AHEAD: all four waves of a workgroup load, compute and wait in step,
// so nothing but the other workgroup of the CU hides a load issued at the top of a step -- these are issued before the MFMA phase of the
// previous step and arrive under it (the other tables are loaded under the products of the first quadratic).
template <bool FOLD>
struct CdtAhead {
    static constexpr int TABLES = FOLD ? 1 : 2;     // a folding round has four entries per table and 238 registers in use: one table ahead (two spill)
    Fr e[FOLD ? 4 : 4];        // per table: FOLD a0, b0, a1, b1 (in[j], in[j + h], in[j + q], in[j + h + q]); else lo, hi
    __device__ __forceinline__ void load(const TablePtrs& tp, size_t j, size_t h, size_t q) {
#pragma unroll
        for (int k = 0; k < TABLES; ++k) {
            if constexpr (FOLD) {
                e[4 * k] = load_fr(tp.in[k], j); e[4 * k + 1] = load_fr(tp.in[k], j + h);
                e[4 * k + 2] = load_fr(tp.in[k], j + q); e[4 * k + 3] = load_fr(tp.in[k], j + h + q);
            } else {
                e[2 * k] = load_fr(tp.in[k], j); e[2 * k + 1] = load_fr(tp.in[k], j + h);
            }
        }
    }
    // (lo, hi) of table k: the entries themselves, or their fold at r written back on the way (round_pair)
    __device__ __forceinline__ void pair(const TablePtrs& tp, int k, size_t j, size_t q, const Fr& r, Fr& lo, Fr& hi) const {
        if constexpr (FOLD) {
            lo = fold_pair(e[4 * k], e[4 * k + 1], r);
            hi = fold_pair(e[4 * k + 2], e[4 * k + 3], r);
            store_fr(tp.out[k], j, lo);
            store_fr(tp.out[k], j + q, hi);
        } else {
            lo = e[2 * k];
            hi = e[2 * k + 1];
        }
    }
};

// the values of output pair j: Q_t (t < K) into plane t of stage_a, (l, d) of the last table into stage_b -- each at this lane's row
template <int K, bool FOLD>
__device__ __forceinline__ void cdt_stage_values(const TablePtrs& tp, const CdtAhead<FOLD>& first, size_t j, size_t h, size_t q, const Fr& r,
                                                 unsigned char* row_a, unsigned char* row_b) {
    static_assert(K == 5, "two quadratics and a linear last factor");
    Fr l0, h0, l1, h1;
    first.pair(tp, 0, j, q, r, l0, h0);
    if constexpr (CdtAhead<FOLD>::TABLES == 2) first.pair(tp, 1, j, q, r, l1, h1);
    else round_pair<FOLD>(tp, 1, j, h, q, r, l1, h1);
    QuadEvals a(l0, h0, l1, h1);
    round_pair<FOLD>(tp, 2, j, h, q, r, l0, h0);
    round_pair<FOLD>(tp, 3, j, h, q, r, l1, h1);
    QuadEvals b(l0, h0, l1, h1);
    round_pair<FOLD>(tp, 4, j, h, q, r, l0, h0);
    cdt_store_row(row_b, l0);
    cdt_store_row(row_b + CDT_PLANE, h0 - l0);
#pragma unroll
    for (int t = 0; t < K; ++t) {
        cdt_store_row(row_a + t * CDT_PLANE, a.cur * b.cur);
        if (t + 1 < K) { a.step(); b.step(); }
    }
}

// The operands of one MFMA step (32 staged indices) of a wave: its own plane of Q, l, d and the fifth plane (used by waves 0 / 1) --
// eight transposed reads issued together, the wait apart (cdt_wait) so that the previous step's MFMAs run under them.  Lane (n = lane % 32,
// h = lane / 32) receives byte n of entries 16 h .. 16 h + 15 of a tile (csm_operands, composed_stage.hpp).
struct CdtOps { csm_u32x2 r[8]; };
__device__ __forceinline__ void cdt_issue(CdtOps& o, uint32_t a_own, uint32_t b_l, uint32_t a_extra) {
    asm volatile(
        "ds_read_b64_tr_b8 %0, %8\n\tds_read_b64_tr_b8 %1, %8 offset:256\n\t"
        "ds_read_b64_tr_b8 %2, %9\n\tds_read_b64_tr_b8 %3, %9 offset:256\n\t"
        "ds_read_b64_tr_b8 %4, %9 offset:%11\n\tds_read_b64_tr_b8 %5, %9 offset:%12\n\t"
        "ds_read_b64_tr_b8 %6, %10\n\tds_read_b64_tr_b8 %7, %10 offset:256"
        : "=&v"(o.r[0]), "=&v"(o.r[1]), "=&v"(o.r[2]), "=&v"(o.r[3]), "=&v"(o.r[4]), "=&v"(o.r[5]), "=&v"(o.r[6]), "=&v"(o.r[7])
        : "v"(a_own), "v"(b_l), "v"(a_extra), "n"(CDT_PLANE), "n"(CDT_PLANE + 256)
        : "memory");
}
__device__ __forceinline__ void cdt_wait(CdtOps& o) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o.r[0]), "+v"(o.r[1]), "+v"(o.r[2]), "+v"(o.r[3]), "+v"(o.r[4]), "+v"(o.r[5]), "+v"(o.r[6]), "+v"(o.r[7])
                 :
                 : "memory");
}
__device__ __forceinline__ mf_v4i cdt_op(const CdtOps& o, int k) {
    mf_v4i x;
    x.x = (int)o.r[2 * k].x; x.y = (int)o.r[2 * k].y; x.z = (int)o.r[2 * k + 1].x; x.w = (int)o.r[2 * k + 1].y;
    return x;
}
// m * x for a small integer m
__device__ __forceinline__ Fr cdt_small_multiple(const Fr& x, uint32_t m) {
    Fr acc = Fr::zero(), p = x;
    for (; m; m >>= 1) { if (m & 1) acc = acc + p; p = p + p; }
    return acc;
}

// Same contract as composed_round_kernel<K, FOLD, false>: the K + 1 sums of this workgroup go to partials[blockIdx.x * rec + rec_off + t];
// FOLD also writes the folded tables.  Dynamic LDS: cdt_lds_bytes(K).
// Accumulator tiles: (plane t, l) and (plane t, d) belong to wave t (t < 4); plane 4's two tiles go to waves 0 (l) and 1 (d).
template <int K, bool FOLD>
static __global__ __launch_bounds__(CDT_ROWS) __attribute__((amdgpu_waves_per_eu(2))) void composed_round_dot_kernel(
    TablePtrs tp, size_t n, const uint64_t* __restrict__ r_ptr, uint32_t rec, uint32_t rec_off, uint64_t* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cdt_lds[];
    constexpr int NP = K;                                  // planes of Q: 0..3 on the wave of that number, 4 shared by waves 0 and 1
    unsigned char* stage_a = cdt_lds;                      // NP planes
    unsigned char* stage_b = cdt_lds + NP * CDT_PLANE;     // l, d
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t h = n >> 1, q = n >> 2, work = FOLD ? q : h;
    const size_t stride = (size_t)gridDim.x * CDT_ROWS;
    const uint32_t iters = (uint32_t)((work + stride - 1) / stride);
    const Fr r = FOLD ? load_fr(r_ptr, 0) : Fr::zero();
    const bool extra = wave < 2;                           // (plane 4, l) on wave 0, (plane 4, d) on wave 1
    mf_v16i acc_l = mf_v16i{0}, acc_d = mf_v16i{0}, acc_x = mf_v16i{0};
    uint32_t ua = 0, ubl = 0, ubd = 0, uax = 0;            // byte sums of the operands this wave reads, per lane
    // operand addresses of this lane inside a staged tile of 32 entries (csm_operands)
    const uint32_t g = lane >> 4, i16 = lane & 15;
    const uint32_t in_tile = 32 * (16 * (g >> 1) + (i16 >> 1)) + 16 * (g & 1) + 8 * (i16 & 1);
    const uint32_t op_a = (uint32_t)(uintptr_t)(stage_a + wave * CDT_PLANE) + in_tile;
    const uint32_t op_b = (uint32_t)(uintptr_t)stage_b + in_tile;
    const uint32_t op_x = (uint32_t)(uintptr_t)(stage_a + 4 * CDT_PLANE) + in_tile;
    CdtAhead<FOLD> ahead;
    {
        const size_t j = (size_t)blockIdx.x * CDT_ROWS + tid;
        if (j < work) ahead.load(tp, j, h, q);
    }
    for (uint32_t it = 0; it < iters; ++it) {
        const size_t j = (size_t)it * stride + (size_t)blockIdx.x * CDT_ROWS + tid;
        unsigned char* row_a = stage_a + tid * 32;
        unsigned char* row_b = stage_b + tid * 32;
        if (j < work) cdt_stage_values<K, FOLD>(tp, ahead, j, h, q, r, row_a, row_b);
        else {
#pragma unroll
            for (int t = 0; t < NP; ++t) cdt_store_row(row_a + t * CDT_PLANE, Fr::zero());
            cdt_store_row(row_b, Fr::zero());
            cdt_store_row(row_b + CDT_PLANE, Fr::zero());
        }
        if (it + 1 < iters && j + stride < work) ahead.load(tp, j + stride, h, q);       // the next step's first two tables, under the MFMA phase
        __syncthreads();                     // the 256 rows are staged
        {
            CdtOps cur, nxt;
            cdt_issue(cur, op_a, op_b, op_x);
#pragma unroll
            for (uint32_t s = 0; s < CDT_ROWS / 32; ++s) {
                cdt_wait(cur);
                if (s + 1 < CDT_ROWS / 32) cdt_issue(nxt, op_a + (s + 1) * 1024, op_b + (s + 1) * 1024, op_x + (s + 1) * 1024);
                const mf_v4i xa = cdt_op(cur, 0), xl = cdt_op(cur, 1), xd = cdt_op(cur, 2);
                ua = csm_bytesum(xa, ua);
                ubl = csm_bytesum(xl, ubl);
                ubd = csm_bytesum(xd, ubd);
                const mf_v4i sa = mfm_signed(xa), sl = mfm_signed(xl), sd = mfm_signed(xd);
                acc_l = __builtin_amdgcn_mfma_i32_32x32x32_i8(sa, sl, acc_l, 0, 0, 0);
                acc_d = __builtin_amdgcn_mfma_i32_32x32x32_i8(sa, sd, acc_d, 0, 0, 0);
                if (extra) {
                    const mf_v4i xx = cdt_op(cur, 3);
                    uax = csm_bytesum(xx, uax);
                    acc_x = __builtin_amdgcn_mfma_i32_32x32x32_i8(mfm_signed(xx), wave == 0 ? sl : sd, acc_x, 0, 0, 0);
                }
                if (s + 1 < CDT_ROWS / 32) cur = nxt;
            }
        }
        __syncthreads();                     // ... and read: the next step may overwrite them
    }
    // ---- epilogue (the staging area is free): per wave and tile the true byte-product sums -> 63 anti-diagonals -> 17 limbs -> REDC
    uint32_t* tmat = reinterpret_cast<uint32_t*>(cdt_lds) + wave * (32 * 33);                                   // [a][b], padded rows
    uint32_t* u_sh = reinterpret_cast<uint32_t*>(cdt_lds + 4 * 32 * 33 * 4) + wave * 128;                       // UA, UB_l, UB_d, UA_extra: [32] each
    unsigned long long* cols = reinterpret_cast<unsigned long long*>(cdt_lds + 4 * 32 * 33 * 4 + 4 * 128 * 4) + wave * (3 * 64);
    Fr* red = reinterpret_cast<Fr*>(cdt_lds + 4 * 32 * 33 * 4 + 4 * 128 * 4 + 4 * 3 * 64 * 8);                  // [plane][l, d]
    const uint32_t col_b = lane & 31, hh = lane >> 5;
    const uint32_t jcount = iters * CDT_ROWS;
    {
        const uint32_t s0 = ua + (uint32_t)__shfl_xor((int)ua, 32, 64), s1 = ubl + (uint32_t)__shfl_xor((int)ubl, 32, 64);
        const uint32_t s2 = ubd + (uint32_t)__shfl_xor((int)ubd, 32, 64), s3 = uax + (uint32_t)__shfl_xor((int)uax, 32, 64);
        if (lane < 32) { u_sh[lane] = s0; u_sh[32 + lane] = s1; u_sh[64 + lane] = s2; u_sh[96 + lane] = s3; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    const int n_tiles = extra ? 3 : 2;                     // of this wave
    for (int u = 0; u < n_tiles; ++u) {
        // tile u: 0 = (own plane, l), 1 = (own plane, d), 2 = (plane 4, l on wave 0 / d on wave 1)
        const uint32_t* ua_row = u == 2 ? u_sh + 96 : u_sh;
        const uint32_t ubb = (u == 0 || (u == 2 && wave == 0)) ? u_sh[32 + col_b] : u_sh[64 + col_b];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const uint32_t a = 8 * (rr >> 2) + 4 * hh + (rr & 3);       // result register rr of lane (b, hh): row a, column b
            const uint32_t gv = (uint32_t)(u == 0 ? acc_l[rr] : u == 1 ? acc_d[rr] : acc_x[rr]);
            tmat[a * 33 + col_b] = gv + 128u * (ua_row[a] + ubb) - 16384u * jcount;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane < 63) {                     // anti-diagonal c = lane
            unsigned long long sacc = 0;
#pragma unroll
            for (int a = 0; a < 32; ++a) {       // (all 32 rows, predicated: the loads are independent of each other and of the running sum)
                const int b = (int)lane - a;
                const uint32_t v = tmat[a * 33 + (b & 31)];
                sacc += (b >= 0 && b < 32) ? v : 0u;
            }
            cols[u * 64 + lane] = sacc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if ((int)lane < n_tiles) {               // lane u: columns -> 17 limbs -> REDC -> the sum of the Montgomery products
        uint32_t x[18];
        unsigned long long carry = 0;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) {
            unsigned long long v = carry;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = 4 * gg + t;
                if (c < 63) v += cols[lane * 64 + c] << (8 * t);     // cols < 2^37: no overflow
            }
            x[gg] = (uint32_t)v;
            carry = v >> 32;
        }
        x[16] = (uint32_t)carry;
        x[17] = 0;
        const uint32_t plane = lane == 2 ? 4u : wave, half = lane == 2 ? wave : lane;
        red[2 * plane + half] = wide_redc(x) * fr_mont_2_32();
    }
    __syncthreads();
    if (tid <= (uint32_t)K) {
        // P_l(t), P_d(t) for t < K from the tiles, at t = K by finite differences (degree K - 1); S(t) = P_l(t) + t P_d(t)
        Fr pl, pd;
        if (tid < (uint32_t)K) { pl = red[2 * tid]; pd = red[2 * tid + 1]; }
        else {                               // P(5) = P(0) - 5 P(1) + 10 P(2) - 10 P(3) + 5 P(4)
            pl = ((red[0] + cdt_small_multiple(red[4], 10)) + cdt_small_multiple(red[8], 5)) - (cdt_small_multiple(red[2], 5) + cdt_small_multiple(red[6], 10));
            pd = ((red[1] + cdt_small_multiple(red[5], 10)) + cdt_small_multiple(red[9], 5)) - (cdt_small_multiple(red[3], 5) + cdt_small_multiple(red[7], 10));
        }
        store_fr(partials, (size_t)blockIdx.x * rec + rec_off + tid, pl + cdt_small_multiple(pd, tid));
    }
}

}  // namespace zk

// mfma_fold.hpp -- the streaming k-variable fold with its limb products on the matrix cores (gfx950).
//
// out[j] = sum_{b < 2^k} w[b] * in[b*m + j]  (Multilinear::partial_evaluation applied k times at variable 0,
// polynomial/src/multilinear/evaluation_form.rs:123-141,162-175; the identity of multifold_kernels.hpp) is HBM work -- 32 bytes
// in per term -- but as VALU code a term costs 64 v_mad_u64_u32 + 64 carry adds, 68 us of issue time at 2^24 against 92 us
// of memory time, and the two only overlap to 107 us.  The products do not have to be VALU work: the
// weights are FIXED for the whole launch, so the schoolbook product of a weight with a table entry, byte by byte,
//     col[c] = sum_i  W[b][c - i] * T[b][j][i],      c = 0..62,  i = 0..31 the entry's bytes as they lie in memory,
// is a matrix product with a Toeplitz matrix of the weight's bytes: A[c][i] = W[b][c - i] (64 x 32 per term), B[i][j] = byte
// i of entry (b, j) (32 x 32 per term and 32 outputs).  v_mfma_i32_32x32x32_i8 does one (32 columns) x (32 outputs) x (one
// term) block per instruction; its B operand is, per lane, 16 consecutive bytes of one table entry -- exactly what a
// global_load_dwordx4 of the table delivers, no transposition, no LDS staging of the table.  The int8 inputs are signed:
//   * the weight is recoded once per workgroup into signed base-256 digits d in [-128, 127] (W + 0x80..80, then ^ 0x80 per byte);
//   * a table byte u is fed as s = u ^ 0x80 = u - 128, and the missing 128 * sum_{b,i} d[b][c - i] is a per-column CONSTANT of
//     the launch (it does not depend on j), added in the epilogue.
// Column sums stay below 2^(19 + k) <= 2^28: exact in the int32 accumulators.  The epilogue (once per 64 outputs x 2^k terms)
// gives every lane one output's 63 columns (v_permlane32_swap), carries them into 17 limbs and runs the same 9-word REDC as
// the VALU form; the weights carry the same factor 2^32, so the result is the same canonical Montgomery residue bit for bit.
// Per term and wave (2 KiB of table): 2 loads, 8 v_xor, 4 ds_read2_b32 (the Toeplitz rows: 16 bytes at a BYTE offset of the term's
// reversed zero-padded digit string of 96 bytes -- kept as four copies shifted by 0..3 bytes so that every lane's window is dword
// aligned: a ds_read_b128 at any address that is not a multiple of 16 takes 64 LDS cycles per wave instead of 8,
// tools/ubench_lds_unaligned.hip, and two of them per term kept the CU's LDS busy 60 % of the time), 4 MFMAs (128 of the SIMD's
// MFMA cycles against ~800 cycles of HBM time for those 2 KiB) -- the VALU is nearly idle and the kernel is bound by memory alone.
// tools/probe_mfma_i8.hip checks the three hardware facts this relies on (unaligned ds_read_b128, operand / result layout of
// the MFMA, v_permlane32_swap); every parity test of the provers and of `evaluation` runs through this kernel.
#pragma once
#include "wide_acc.hpp"

namespace zk {

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

constexpr int MFM_CHUNK = 128;      // terms whose digit strings sit in LDS at a time (4 x 12 KiB)
constexpr int MFM_QSTRIDE = 96;     // bytes per term: Q[y] = d[63 - y] for y in [32, 64), zero elsewhere
// LDS: four planes, plane a = the strings of all terms shifted down by a bytes (dword d of a term = Q[4 d + a .. 4 d + a + 3]);
// a plane is 24 dwords per term + 16, so the windows a wave-instruction touches in the four planes fall into different banks
__host__ __device__ constexpr uint32_t mfm_plane_dwords(uint32_t chunk) { return chunk * 24 + 16; }
__host__ __device__ constexpr size_t mfm_lds_bytes(uint32_t chunk) { return (size_t)16 * mfm_plane_dwords(chunk); }

// raw 16 bytes of a table entry; the ^ 0x80 happens at the point of use (applied at the load it made the compiler wait for
// every load at once: no loads in flight)
__device__ __forceinline__ mf_v4i mfm_load_nt(const unsigned char* p) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    return __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(p));
}
__device__ __forceinline__ mf_v4i mfm_signed(mf_v4i a) {
    a.x ^= (int)0x80808080u; a.y ^= (int)0x80808080u; a.z ^= (int)0x80808080u; a.w ^= (int)0x80808080u;
    return a;
}

// One wave folds one tile of 64 outputs: lanes (n = lane % 32, h = lane / 32) load bytes [16 h, 16 h + 16) of entries j0 + n
// (jh = 0) and j0 + 32 + n (jh = 1) of every term; acc[jh][mh] holds columns 32 mh .. 32 mh + 31 of those 32 outputs.  Two register
// sets of DEPTH terms are in flight (16 loads of 1 KiB per wave at DEPTH = 4).
// ORDER OF THE TERMS.  The sum is exact, so a wave may take its terms in any order, and the order decides the memory rate: with
// every wave starting at term 0 the chip sweeps one row of the table (n / 2^k entries: 8 MiB at k = 6) at a time and reaches
// 5.9-6.1 TB/s on these loads alone; with tile T starting at term T mod 2^k the waves are spread over all rows at every moment and
// the same loads run at 6.8 TB/s at any occupancy from 4 to 16 waves per CU (tools/ubench_rows.hip, profiles/r03/ubench_rows.txt;
// several tiles per wave as one long stream measured 4.6-5.4 TB/s either way, so a wave takes ONE tile).
// Host contract: 2^k is a multiple of 2 DEPTH.
// WSUM (MultilinearTrait::evaluation with every point known, evaluation_form.rs:162-175): the folded table is not written; tile T's
// record is sum_j eq[j] * out[j] over its 64 outputs -- with eq = the eq table of the REMAINING points (kept as the two factors of
// an outer product, eval_weights_kernel) the records add up to the evaluation, so the whole evaluation is this one pass over the
// table: 32 n bytes read, nothing written but m / 64 records.
template <int WAVES, int DEPTH, bool WSUM = false>
static __global__ __launch_bounds__(64 * WAVES) void multifold_mfma_kernel(const uint64_t* __restrict__ in, size_t m, uint32_t k,
                                                                          const uint64_t* __restrict__ weights,
                                                                          uint64_t* __restrict__ out,
                                                                          uint64_t* __restrict__ partials, uint32_t rot,
                                                                          const uint64_t* __restrict__ out_wa = nullptr,
                                                                          const uint64_t* __restrict__ out_wb = nullptr, uint32_t out_s = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];   // mfm_lds_bytes(min(2^k, MFM_CHUNK)) (host)
    uint32_t* qd = reinterpret_cast<uint32_t*>(zk_dyn_lds);
    const unsigned char* q_lds = zk_dyn_lds;                        // plane 0 = the unshifted strings
    __shared__ int32_t d_lds[32];        // digit sums of the current chunk
    __shared__ int32_t kc_lds[64];       // 128 * sum_{b,i} d[b][c - i], all chunks
    __shared__ long long kv_lds[16];     // the same per 32-bit limb: sum_t kc[4 g + t] << 8 t
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = lane & 31, h = lane >> 5;
    const uint32_t n_terms = 1u << k;
    const size_t tile = (size_t)blockIdx.x * WAVES + wave;
    const size_t row = 32 * m;                                       // bytes between consecutive terms of one output
    const unsigned char* p = reinterpret_cast<const unsigned char*>(in) + 32 * (tile * 64 + n) + 16 * h;
    const uint32_t chunk = n_terms < (uint32_t)MFM_CHUNK ? n_terms : (uint32_t)MFM_CHUNK;   // terms per chunk, a power of two
    const uint32_t plane = mfm_plane_dwords(chunk);
    // this lane's window of a term's string starts at byte s = 31 - n + 16 h (columns 32..63, mh = 1) and s + 32 (columns 0..31):
    // plane s % 4, dword s / 4 (+ 8 for mh = 0)
    const uint32_t* qa = qd + ((31 - n + 16 * h) & 3) * plane + ((31 - n + 16 * h) >> 2);
    const uint32_t start = (uint32_t)(tile * rot) & (chunk - 1);     // this tile's first term inside every chunk
    const uint32_t last = n_terms - 1;
    // step s of the wave's stream: chunk s / chunk, term ((s % chunk) + start) % chunk of it; steps past the end are clamped
    // to the last one and never used
    // u-th term this tile takes inside a chunk.  (Measured and dropped: the upper half of the table -- what the sums pass read
    // last and the Infinity Cache may still hold -- first, rotated inside each half: 104 us against 93-96 on the same box.)
    auto term_in_chunk = [&](uint32_t u) -> uint32_t { return (u + start) & (chunk - 1); };
    auto term_of = [&](uint32_t s) -> uint32_t {
        s = s < last ? s : last;
        return (s & ~(chunk - 1)) | term_in_chunk(s & (chunk - 1));
    };
    mf_v4i da[DEPTH][2], db[DEPTH][2];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) { const unsigned char* pt = p + (size_t)term_of(u) * row; da[u][0] = mfm_load_nt(pt); da[u][1] = mfm_load_nt(pt + 1024); }
    if (tid < 64) kc_lds[tid] = 0;
    mf_v16i acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};    // acc[jh][mh]

    for (uint32_t t0 = 0; t0 < n_terms; t0 += chunk) {
        __syncthreads();                                             // the previous chunk's strings are no longer read
        for (uint32_t idx = tid; idx < 4 * chunk; idx += 64 * WAVES) {
            // signed digits: S = W + 0x80...80 (no overflow: W < r < 0.46 * 2^256), digit bytes = S ^ 0x80...80
            const uint32_t t = idx >> 2, a = idx & 3;
            const Fr w = load_fr(weights, t0 + t);
            uint32_t qs[25];                                         // the string as dwords: 8 zero, 8 of digits (reversed), 8 zero, and one more
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { qs[i] = 0; qs[16 + i] = 0; }
            qs[24] = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) { c += (uint64_t)w.l[i] + 0x80808080u; qs[15 - i] = __builtin_bswap32((uint32_t)c ^ 0x80808080u); c >>= 32; }
            uint32_t* q = qd + a * plane + t * 24;
#pragma unroll
            for (int d = 0; d < 24; ++d) q[d] = (uint32_t)((((uint64_t)qs[d + 1] << 32) | qs[d]) >> (8 * a));
        }
        __syncthreads();
        if (tid < 32) {                                              // digit x of every term of the chunk: Q[t][63 - x]
            int32_t dsum = 0;
            for (uint32_t t = 0; t < chunk; ++t) dsum += (int32_t)(signed char)q_lds[t * MFM_QSTRIDE + 63 - tid];
            d_lds[tid] = dsum;
        }
        __syncthreads();
        if (tid < 64) {                                              // column c collects digits c - 31 .. c
            int32_t s = 0;
            const int lo = (int)tid - 31 < 0 ? 0 : (int)tid - 31, hi = tid < 31 ? (int)tid : 31;
            for (int x = lo; x <= hi; ++x) s += d_lds[x];
            kc_lds[tid] += 128 * s;
        }
        // ---- the chunk's terms, DEPTH at a time, two register sets
        for (uint32_t t = 0; t < chunk; t += 2 * DEPTH) {
            const uint32_t g = t0 + t;                                   // step index of da[0]
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) { const unsigned char* pt = p + (size_t)term_of(g + DEPTH + u) * row; db[u][0] = mfm_load_nt(pt); db[u][1] = mfm_load_nt(pt + 1024); }
            __builtin_amdgcn_sched_barrier(0);                       // keep the issue order: the scheduler sank these loads below the MFMAs they should overlap
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const uint32_t* qt = qa + term_in_chunk(t + u) * 24;
                const mf_v4i a0 = {(int)qt[8], (int)qt[9], (int)qt[10], (int)qt[11]}, a1 = {(int)qt[0], (int)qt[1], (int)qt[2], (int)qt[3]};
                const mf_v4i b0 = mfm_signed(da[u][0]), b1 = mfm_signed(da[u][1]);
                acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc11, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) { const unsigned char* pt = p + (size_t)term_of(g + 2 * DEPTH + u) * row; da[u][0] = mfm_load_nt(pt); da[u][1] = mfm_load_nt(pt + 1024); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const uint32_t* qt = qa + term_in_chunk(t + DEPTH + u) * 24;
                const mf_v4i a0 = {(int)qt[8], (int)qt[9], (int)qt[10], (int)qt[11]}, a1 = {(int)qt[0], (int)qt[1], (int)qt[2], (int)qt[3]};
                const mf_v4i b0 = mfm_signed(db[u][0]), b1 = mfm_signed(db[u][1]);
                acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc11, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    if (tid < 16) {
        long long v = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) v += (long long)kc_lds[4 * tid + t] << (8 * t);
        kv_lds[tid] = v;
    }
    __syncthreads();
    // ---- epilogue: lane (n, h) takes output 32 h + n.  Result register r = 4 q + t of acc[jh][mh] holds column
    // 32 mh + 8 q + 4 h + t of output 32 jh + n; the swap hands the upper lanes' jh = 0 rows to the lower lanes and the lower
    // lanes' jh = 1 rows to the upper ones: afterwards acc0* holds the h' = 0 rows and acc1* the h' = 1 rows of MY output
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)acc00[r], (unsigned)acc10[r], false, false);
        acc00[r] = (int)s0[0]; acc10[r] = (int)s0[1];
        auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)acc01[r], (unsigned)acc11[r], false, false);
        acc01[r] = (int)s1[0]; acc11[r] = (int)s1[1];
    }
    uint32_t x[18];
    long long carry = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {                                   // limb g = columns 4 g .. 4 g + 3 = (mh, q, h') = (g / 8, (g / 2) % 4, g % 2)
        const int mh = g >> 3, q = (g >> 1) & 3, hp = g & 1;
        long long v = carry + kv_lds[g];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = 4 * q + t;
            const int col = hp ? (mh ? acc11[r] : acc10[r]) : (mh ? acc01[r] : acc00[r]);
            v += (long long)col << (8 * t);
        }
        x[g] = (uint32_t)v;
        carry = v >> 32;
    }
    x[16] = (uint32_t)carry;
    x[17] = 0;
    Fr o = wide_redc(x);
    if constexpr (WSUM) {                                            // eq[j] of the remaining points = wa[j >> s] * wb[j & (2^s - 1)]
        const size_t j = tile * 64 + lane;
        o = o * (load_fr(out_wa, j >> out_s) * load_fr(out_wb, j & (((size_t)1 << out_s) - 1)));
    }
    else store_fr(out, tile * 64 + lane, o);
    const Fr s = wave_reduce_fr(o);
    if (lane == 0) store_fr(partials, tile, s);
}

}  // namespace zk

// composed_kernels.hpp -- composed / multi-composed sumcheck prover kernels for gfx950.
//
// Replaces the round loops of
//   ComposedSumcheck::prove                       sumcheck/src/composed/composed_sumcheck.rs:32-67
//   MultiComposedSumcheckProver::prove_internal   sumcheck/src/composed/multi_composed_sumcheck.rs:64-121
// and, inside them, ComposedMultilinear::{partial_evaluation, element_wise_product}
// (polynomial/src/composed/composed_multilinear.rs:63-75,105-111).
//
// The reference evaluates a round polynomial by folding every table at t = 0..=K ((K+1)*K full folds),
// multiplying them element-wise and summing.  Here one pass over the tables does it all: a table's value
// at integer t is lo + t*(hi - lo), obtained by repeated addition of d = hi - lo (no multiplication), the
// K-way product costs K-1 Montgomery products per t, and -- from the second round on -- the same pass
// first folds the tables at the previous challenge and writes them back (fused, like the basic prover).
// Algorithmic traffic per round and table: read 32 n + write 16 n bytes (n = entries before the fold).
#pragma once
#include "mle_kernels.hpp"

namespace zk {

constexpr int CMP_MAX_K = 5;        // tables per product term (K <= 5)
constexpr int CMP_MAX_TERMS = 4;    // product terms (GKR uses 2)
constexpr int CMP_MAX_REC = 16;     // sums per workgroup record, sum_p (K_p + 1)
constexpr int CMP_MAX_MONO = 7;     // monomials of a round polynomial (degree <= 6)

struct TablePtrs {
    const uint64_t* in[CMP_MAX_K];
    uint64_t* out[CMP_MAX_K];
};

// evaluations at t = 0..K of prod_k (lo_k + t*(hi_k - lo_k)), added into sums[0..K]
template <int K>
__device__ __forceinline__ void accumulate_round_evals(const Fr (&lo)[K], const Fr (&hi)[K], Fr (&sums)[K + 1]) {
    Fr v[K], d[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = lo[k]; d[k] = hi[k] - lo[k]; }
#pragma unroll
    for (int t = 0; t <= K; ++t) {
        Fr prod = v[0];
#pragma unroll
        for (int k = 1; k < K; ++k) prod = prod * v[k];
        sums[t] = sums[t] + prod;
        if (t < K) {
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = v[k] + d[k];
        }
    }
}

// One round of one product term.
//   FOLD = false: evaluate the round polynomial of the tables as they are (first round): pairs (j, j + n/2).
//   FOLD = true : fold every table at *r_ptr first (in[j], in[j + n/2] -> out[j]) for the two outputs j and
//                 j + n/4, store them, and evaluate the NEXT round polynomial on that folded pair.
// Per workgroup, the K+1 sums go to partials[(block * rec + rec_off + t)].
template <int K, bool FOLD>
static __global__ __launch_bounds__(MLE_BLOCK) void composed_round_kernel(TablePtrs tp, size_t n, const uint64_t* __restrict__ r_ptr,
                                                                   uint32_t rec, uint32_t rec_off,
                                                                   uint64_t* __restrict__ partials) {
    __shared__ Fr red[MLE_BLOCK / 64];
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    Fr sums[K + 1];
#pragma unroll
    for (int t = 0; t <= K; ++t) sums[t] = Fr::zero();
    if (FOLD) {
        const Fr r = load_fr(r_ptr, 0);
        const size_t h = n >> 1, q = n >> 2;   // q = pairs of the folded table (may be 0 when n == 2)
        if (q == 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) store_fr(tp.out[k], 0, fold_pair(load_fr(tp.in[k], 0), load_fr(tp.in[k], 1), r));
            }
        }
        for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < q; j += stride) {
            Fr lo[K], hi[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                Fr a0 = load_fr(tp.in[k], j), a1 = load_fr(tp.in[k], j + q);
                Fr b0 = load_fr(tp.in[k], j + h), b1 = load_fr(tp.in[k], j + h + q);
                lo[k] = fold_pair(a0, b0, r);
                hi[k] = fold_pair(a1, b1, r);
                store_fr(tp.out[k], j, lo[k]);
                store_fr(tp.out[k], j + q, hi[k]);
            }
            accumulate_round_evals<K>(lo, hi, sums);
        }
    } else {
        const size_t h = n >> 1;
        for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < h; j += stride) {
            Fr lo[K], hi[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { lo[k] = load_fr(tp.in[k], j); hi[k] = load_fr(tp.in[k], j + h); }
            accumulate_round_evals<K>(lo, hi, sums);
        }
    }
#pragma unroll
    for (int t = 0; t <= K; ++t) {
        Fr s = block_reduce_fr(sums[t], red);
        if (threadIdx.x == 0) store_fr(partials, (size_t)blockIdx.x * rec + rec_off + t, s);
    }
}

// sum_j prod_k table_k[j]  (ComposedSumcheck::calculate_poly_sum composed_sumcheck.rs:28-30): one partial per workgroup
template <int K>
static __global__ __launch_bounds__(MLE_BLOCK) void product_sum_kernel(TablePtrs tp, size_t n, uint64_t* __restrict__ partials) {
    __shared__ Fr red[MLE_BLOCK / 64];
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    Fr s = Fr::zero();
    for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < n; j += stride) {
        Fr prod = load_fr(tp.in[0], j);
#pragma unroll
        for (int k = 1; k < K; ++k) prod = prod * load_fr(tp.in[k], j);
        s = s + prod;
    }
    s = block_reduce_fr(s, red);
    if (threadIdx.x == 0) store_fr(partials, blockIdx.x, s);
}
// out[0] = sum of n_partials values (+ *accumulate_into if given)
static __global__ __launch_bounds__(MLE_BLOCK) void finish_sum_kernel(const uint64_t* __restrict__ partials, uint32_t n_partials,
                                                               uint64_t* __restrict__ out, uint32_t accumulate) {
    __shared__ Fr red[MLE_BLOCK / 64];
    Fr s = Fr::zero();
    for (uint32_t b = threadIdx.x; b < n_partials; b += MLE_BLOCK) s = s + load_fr(partials, b);
    s = block_reduce_fr(s, red);
    if (threadIdx.x == 0) store_fr(out, 0, accumulate ? s + load_fr(out, 0) : s);
}

// ---- per-round control kernel -------------------------------------------------------------------------
struct ComposedMeta {
    uint32_t n_terms;
    uint32_t k[CMP_MAX_TERMS];         // tables per term (= degree of the term's round polynomial)
    uint32_t rec_off[CMP_MAX_TERMS];   // offset of the term's K+1 sums inside a workgroup record
    uint32_t rec;                      // sums per record
    uint32_t multi;                    // 0: ComposedSumcheck transcript (raw evaluations), 1: multi-composed (sparse coefficients)
};
// Device-resident state.  interp[d] is the (d+1)x(d+1) matrix taking evaluations at x = 0..d to coefficients
// (Montgomery form), uploaded by the host once per prove.
struct ComposedDev {
    Sha256State transcript;
    uint64_t sum[4];
    uint64_t interp[CMP_MAX_K + 1][(CMP_MAX_K + 1) * (CMP_MAX_K + 1)][4];
};

// Closes a round: reduce the workgroup records, build the round polynomial, absorb it, derive the challenge.
//   first: 1 = the transcript is started here (ComposedSumcheck: nothing absorbed before, composed_sumcheck.rs:33;
//              multi-composed prove_partial: the claimed sum, multi_composed_sumcheck.rs:60,70),
//          2 = the transcript state was prepared by the host (multi-composed `prove`: all table bytes were
//              hashed first, :51-53) and the claimed sum is absorbed here,
//          0 = continue.
// Output per round (round_out + 64 * round, in u64):
//   multi == 0: K+1 evaluations (4 u64 each);  multi == 1: [0] = #monomials, then (coeff, pow) pairs of 8 u64 from [8].
static __global__ __launch_bounds__(MLE_BLOCK) void composed_transcript_kernel(const uint64_t* __restrict__ partials,
                                                                        uint32_t n_partials, ComposedMeta meta,
                                                                        ComposedDev* st, uint32_t round, uint32_t first,
                                                                        uint64_t* __restrict__ round_out,
                                                                        uint64_t* __restrict__ challenges) {
    __shared__ Fr red[MLE_BLOCK / 64];
    __shared__ Fr evals[CMP_MAX_REC];
    for (uint32_t v = 0; v < meta.rec; ++v) {
        Fr s = Fr::zero();
        for (uint32_t b = threadIdx.x; b < n_partials; b += MLE_BLOCK) s = s + load_fr(partials, (size_t)b * meta.rec + v);
        s = block_reduce_fr(s, red);
        if (threadIdx.x == 0) evals[v] = s;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    Transcript tr;
    if (first == 1) tr.init(); else tr.load(&st->transcript);
    uint64_t* out = round_out + 64 * (size_t)round;
    if (!meta.multi) {
        // transcript.commit(&vec_to_bytes(&round_poly))  composed_sumcheck.rs:51
        for (uint32_t t = 0; t <= meta.k[0]; ++t) {
            tr.commit_fr(evals[t]);
            store_fr(out, t, evals[t]);
        }
    } else {
        if (first) tr.commit_fr(load_fr(st->sum, 0));   // multi_composed_sumcheck.rs:70
        // round_poly = sum over terms of interpolation(evals at x = 0..K)  (:79-95); coefficients that are zero
        // are dropped per term (sparse_univariate.rs:55) but a zero produced by the sum is kept (:159-203).
        Fr coeff[CMP_MAX_MONO];
        bool present[CMP_MAX_MONO];
        for (int k = 0; k < CMP_MAX_MONO; ++k) { coeff[k] = Fr::zero(); present[k] = false; }
        for (uint32_t p = 0; p < meta.n_terms; ++p) {
            const uint32_t d = meta.k[p];
            for (uint32_t k = 0; k <= d; ++k) {
                Fr c = Fr::zero();
                for (uint32_t i = 0; i <= d; ++i)
                    c = c + fr_mul_outlined(load_fr(&st->interp[d][k * (d + 1) + i][0], 0), evals[meta.rec_off[p] + i]);
                if (!c.is_zero()) { coeff[k] = coeff[k] + c; present[k] = true; }
            }
        }
        uint32_t n_mono = 0;
        for (uint32_t k = 0; k < CMP_MAX_MONO; ++k) {
            if (!present[k]) continue;
            tr.commit_fr(coeff[k]);                 // coeff || pow, 32 bytes big-endian each (sparse_univariate.rs:27-34)
            uint32_t pw[8] = {0, 0, 0, 0, 0, 0, 0, k};
            tr.commit_words8(pw);
            store_fr(out + 8, 2 * n_mono, coeff[k]);
            Fr powm = Fr::zero();
            powm.l[0] = k;
            store_fr(out + 8, 2 * n_mono + 1, fr_to_mont_outlined(powm));
            ++n_mono;
        }
        out[0] = n_mono;
    }
    Fr r = tr.challenge_fr();
    tr.store(&st->transcript);
    store_fr(challenges, round, r);
}

}  // namespace zk

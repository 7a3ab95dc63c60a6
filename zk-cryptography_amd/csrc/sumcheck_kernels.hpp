// sumcheck_kernels.hpp -- the basic sumcheck prover's per-round control kernels for gfx950.
//
// Replaces the round loop of Sumcheck::prove (sumcheck/src/sumcheck.rs:29-61).  The
// data-parallel part of a round (half-sums + fold) is mle_kernels.hpp's fold_kernel<true>;
// the kernels here close a round: reduce the per-workgroup partial sums, absorb the round
// polynomial into the device-resident transcript and derive the next challenge -- so the
// strictly sequential Fiat-Shamir chain never crosses PCIe.
#pragma once
#include "mle_kernels.hpp"

namespace zk {

// Device-resident prover bookkeeping (one per context).
struct SumcheckDev {
    Sha256State transcript;
    uint64_t sum[4];   // claimed sum (Montgomery)
};

// Closes round `round` while the table is still large.
//   partials: n_partials x (lo, hi) partial half-sums of the CURRENT table (from half_sums_kernel or the
//             previous round's fold_kernel<true>); for world > 1 they are the all-gathered per-rank sums.
//   first:    round 0 also starts the transcript and absorbs the claimed sum (sumcheck.rs:31-35):
//             1 = the sum is lo + hi (computed here), 2 = the caller's `self.sum`, already in st->sum.
// Writes round_polys[round] = (lo, hi) and challenges[round] (both Montgomery form).
__global__ __launch_bounds__(MLE_BLOCK) void sumcheck_round_kernel(const uint64_t* __restrict__ partials,
                                                                   uint32_t n_partials, SumcheckDev* st,
                                                                   uint32_t round, uint32_t first,
                                                                   uint64_t* __restrict__ round_polys,
                                                                   uint64_t* __restrict__ challenges) {
    __shared__ Fr red[MLE_BLOCK / 64];
    Fr lo, hi;
    reduce_partials(partials, n_partials, red, lo, hi);
    if (threadIdx.x == 0) {
        Transcript tr;
        if (first) {
            Fr sum = (first == 2) ? load_fr(st->sum, 0) : lo + hi;
            store_fr(st->sum, 0, sum);
            tr.init();
            tr.commit_fr(sum);
        } else {
            tr.load(&st->transcript);
        }
        tr.commit_fr(lo);   // uni_poly.to_bytes()  sumcheck.rs:42
        tr.commit_fr(hi);
        Fr r = tr.challenge_fr();   // :46
        tr.store(&st->transcript);
        store_fr(round_polys, 2 * (size_t)round, lo);
        store_fr(round_polys, 2 * (size_t)round + 1, hi);
        store_fr(challenges, round, r);
    }
}

// Runs ALL remaining rounds of a table of n <= TAIL_N entries inside one workgroup (table in LDS):
// per round half-sums -> transcript -> challenge -> fold (sumcheck.rs:40-51).  `first` as above.
__global__ __launch_bounds__(MLE_BLOCK) void sumcheck_tail_kernel(const uint64_t* __restrict__ in, uint32_t n,
                                                                  SumcheckDev* st, uint32_t round0, uint32_t first,
                                                                  uint64_t* __restrict__ round_polys,
                                                                  uint64_t* __restrict__ challenges,
                                                                  uint64_t* __restrict__ final_eval) {
    __shared__ Fr tab[TAIL_N];
    __shared__ Fr red[MLE_BLOCK / 64];
    __shared__ Fr r_sh;
    for (uint32_t j = threadIdx.x; j < n; j += MLE_BLOCK) tab[j] = load_fr(in, j);
    __syncthreads();
    Transcript tr;
    if (threadIdx.x == 0 && !first) tr.load(&st->transcript);
    uint32_t cur = n, round = round0;
    while (cur > 1) {
        const uint32_t half = cur >> 1;
        Fr s_lo = Fr::zero(), s_hi = Fr::zero();
        for (uint32_t j = threadIdx.x; j < half; j += MLE_BLOCK) {
            s_lo = s_lo + tab[j];
            s_hi = s_hi + tab[j + half];
        }
        Fr lo = block_reduce_fr(s_lo, red);
        Fr hi = block_reduce_fr(s_hi, red);
        if (threadIdx.x == 0) {
            if (first && round == round0) {
                Fr sum = (first == 2) ? load_fr(st->sum, 0) : lo + hi;
                store_fr(st->sum, 0, sum);
                tr.init();
                tr.commit_fr(sum);
            }
            tr.commit_fr(lo);
            tr.commit_fr(hi);
            Fr r = tr.challenge_fr();
            r_sh = r;
            store_fr(round_polys, 2 * (size_t)round, lo);
            store_fr(round_polys, 2 * (size_t)round + 1, hi);
            store_fr(challenges, round, r);
        }
        __syncthreads();
        const Fr r = r_sh;
        Fr o[TAIL_N / 2 / MLE_BLOCK];
#pragma unroll
        for (int u = 0; u < TAIL_N / 2 / MLE_BLOCK; ++u) {
            uint32_t j = threadIdx.x + u * MLE_BLOCK;
            if (j < half) o[u] = fold_pair(tab[j], tab[j + half], r);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < TAIL_N / 2 / MLE_BLOCK; ++u) {
            uint32_t j = threadIdx.x + u * MLE_BLOCK;
            if (j < half) tab[j] = o[u];
        }
        __syncthreads();
        cur = half;
        ++round;
    }
    if (threadIdx.x == 0) {
        tr.store(&st->transcript);
        store_fr(final_eval, 0, tab[0]);
    }
}

}  // namespace zk

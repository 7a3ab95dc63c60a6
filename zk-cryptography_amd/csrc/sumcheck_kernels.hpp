// sumcheck_kernels.hpp -- the basic sumcheck prover's per-round control kernels for gfx950.
//
// Replaces the round loop of Sumcheck::prove (sumcheck/src/sumcheck.rs:29-61).  The
// data-parallel part of a round (half-sums + fold) is mle_kernels.hpp's fold_kernel<true>;
// the kernels here close a round: reduce the per-workgroup partial sums, absorb the round
// polynomial into the device-resident transcript and derive the next challenge -- so the
// strictly sequential Fiat-Shamir chain never crosses PCIe.
#pragma once
#include "mle_kernels.hpp"

#include "stamps.hpp"

namespace zk {


// Device-resident prover bookkeeping (one per context).
struct SumcheckDev {
    Sha256State transcript;
    uint64_t sum[4];   // claimed sum (Montgomery)
};

// One round of the transcript (sumcheck.rs:33-35,42-46) executed by wave 0 of the workgroup: lanes 0..2 convert
// (sum, lo, hi) out of Montgomery form side by side -- one product's latency instead of three -- then every lane
// of the wave runs the same SHA-256 chain (uniform control flow) and lane 0 publishes.  `conv` is LDS scratch.
__device__ __forceinline__ Fr transcript_round(Transcript& tr, Fr* conv, const Fr& sum, const Fr& lo, const Fr& hi,
                                               bool absorb_sum) {
    const int lane = threadIdx.x & 63;
    if (lane == 0) { conv[0] = sum; conv[1] = lo; conv[2] = hi; }
    __builtin_amdgcn_wave_barrier();
    Fr mine = conv[lane < 3 ? lane : 2];
    Fr canon = fr_from_mont_outlined(mine);
    Fr c_sum, c_lo, c_hi;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) {
        c_sum.l[i] = __shfl(canon.l[i], 0, 64);
        c_lo.l[i] = __shfl(canon.l[i], 1, 64);
        c_hi.l[i] = __shfl(canon.l[i], 2, 64);
    }
    if (absorb_sum) tr.commit_canonical(c_sum);
    tr.commit_canonical(c_lo);
    tr.commit_canonical(c_hi);
    return tr.challenge_fr();
}

// Closes round `round` while the table is still large.
//   partials: n_partials x (lo, hi) partial half-sums of the CURRENT table (from half_sums_kernel or the
//             previous round's fold_kernel<true>); for world > 1 they are the all-gathered per-rank sums.
//   first:    round 0 also starts the transcript and absorbs the claimed sum (sumcheck.rs:31-35):
//             1 = the sum is lo + hi (computed here), 2 = the caller's `self.sum`, passed by value (claimed).
//   have_hs:  the first round's (lo, hi) were already computed by poly_sum() and arrive by value (hs_lo, hs_hi).
// Writes round_polys[round] = (lo, hi) and challenges[round] (both Montgomery form).
static __global__ __launch_bounds__(MLE_BLOCK) void sumcheck_round_kernel(const uint64_t* __restrict__ partials,
                                                                   uint32_t n_partials, SumcheckDev* st,
                                                                   uint32_t round, uint32_t first, FrArg claimed,
                                                                   uint32_t have_hs, FrArg hs_lo, FrArg hs_hi,
                                                                   uint64_t* __restrict__ round_polys,
                                                                   uint64_t* __restrict__ challenges) {
    __shared__ Fr red[2 * MLE_BLOCK / 64];
    __shared__ Fr conv[4];
    Fr lo, hi;
    ZK_STAMP(0);
    if (have_hs) { lo = fr_from_arg(hs_lo); hi = fr_from_arg(hs_hi); }
    else reduce_partials(partials, n_partials, red, lo, hi);
    ZK_STAMP(1);
    if (threadIdx.x == 0) { conv[1] = lo; conv[2] = hi; }
    __syncthreads();
    if (threadIdx.x < 64) {   // wave 0, uniform
        lo = conv[1];
        hi = conv[2];
        Transcript tr;
        Fr sum = Fr::zero();
        if (first) {
            sum = (first == 2) ? fr_from_arg(claimed) : lo + hi;
            tr.init();
        } else {
            tr.load(&st->transcript);
        }
        ZK_STAMP(2);
        Fr r = transcript_round(tr, conv, sum, lo, hi, first != 0);   // sumcheck.rs:33-35,42,46
        ZK_STAMP(3);
        if (threadIdx.x == 0) {
            if (first) store_fr(st->sum, 0, sum);
            tr.store(&st->transcript);
            store_fr(round_polys, 2 * (size_t)round, lo);
            store_fr(round_polys, 2 * (size_t)round + 1, hi);
            store_fr(challenges, round, r);
        }
    }
}

// Runs ALL remaining rounds of a table of n <= TAIL_N entries inside one workgroup (table in LDS):
// per round half-sums -> transcript -> challenge -> fold (sumcheck.rs:40-51).  `first` as above.
static __global__ __launch_bounds__(MLE_BLOCK) void sumcheck_tail_kernel(const uint64_t* __restrict__ in, uint32_t n,
                                                                  SumcheckDev* st, uint32_t round0, uint32_t first,
                                                                  FrArg claimed, uint64_t* __restrict__ round_polys,
                                                                  uint64_t* __restrict__ challenges,
                                                                  uint64_t* __restrict__ final_eval) {
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    Fr* tab = reinterpret_cast<Fr*>(zk_dyn_lds);   // TAIL_N entries
    Fr* red = tab + TAIL_N;                        // 2 * MLE_BLOCK / 64
    Fr* conv = red + 2 * MLE_BLOCK / 64;           // 4
    Fr* r_shp = conv + 4;                          // 1
#define r_sh (*r_shp)
    for (uint32_t j = threadIdx.x; j < n; j += MLE_BLOCK) tab[j] = load_fr(in, j);
    __syncthreads();
    Transcript tr;
    if (threadIdx.x < 64 && !first) tr.load(&st->transcript);
    uint32_t cur = n, round = round0;
    while (cur > 1) {
        const uint32_t half = cur >> 1;
        Fr s_lo = Fr::zero(), s_hi = Fr::zero();
        for (uint32_t j = threadIdx.x; j < half; j += MLE_BLOCK) {
            s_lo = s_lo + tab[j];
            s_hi = s_hi + tab[j + half];
        }
        Fr lo = s_lo, hi = s_hi;
        block_reduce_fr2(lo, hi, red);
        if (threadIdx.x == 0) { conv[1] = lo; conv[2] = hi; }
        __syncthreads();
        if (threadIdx.x < 64) {   // wave 0, uniform
            lo = conv[1];
            hi = conv[2];
            const bool absorb_sum = first && round == round0;
            Fr sum = Fr::zero();
            if (absorb_sum) {
                sum = (first == 2) ? fr_from_arg(claimed) : lo + hi;
                tr.init();
            }
            Fr r = transcript_round(tr, conv, sum, lo, hi, absorb_sum);
            if (threadIdx.x == 0) {
                if (absorb_sum) store_fr(st->sum, 0, sum);
                r_sh = r;
                store_fr(round_polys, 2 * (size_t)round, lo);
                store_fr(round_polys, 2 * (size_t)round + 1, hi);
                store_fr(challenges, round, r);
            }
        }
        __syncthreads();
        const Fr r = r_sh;
        // in place: lane j reads (j, j+half) and writes j; no other lane touches index j this round
        for (uint32_t j = threadIdx.x; j < half; j += MLE_BLOCK) tab[j] = fold_pair(tab[j], tab[j + half], r);
        __syncthreads();
        cur = half;
        ++round;
    }
    if (threadIdx.x == 0) {
        tr.store(&st->transcript);
        store_fr(final_eval, 0, tab[0]);
    }
#undef r_sh
}
constexpr size_t TAIL_LDS_BYTES = (size_t)(TAIL_N + 2 * MLE_BLOCK / 64 + 4 + 1) * sizeof(Fr);

}  // namespace zk

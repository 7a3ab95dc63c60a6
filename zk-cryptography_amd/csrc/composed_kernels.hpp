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
#include "stamps.hpp"
#include "wide_acc.hpp"
#include "outer_transcript.hpp"

namespace zk {

constexpr int CMP_MAX_K = 5;        // tables per product term (K <= 5)
constexpr int CMP_MAX_TERMS = 4;    // product terms (GKR uses 2)
constexpr int CMP_MAX_REC = 16;     // sums per workgroup record, sum_p (K_p + 1)
constexpr int CMP_MAX_MONO = 7;     // monomials of a round polynomial (degree <= 6)
constexpr int CMP_MAX_BLOCKS = 9;   // SHA-256 blocks of one round: <= 15 pending words + sum (8) + 7 x 16 + padding

struct TablePtrs {
    const uint64_t* in[CMP_MAX_K];
    uint64_t* out[CMP_MAX_K];
    // optional additive table of the term (term = prod_k table_k + lin): not a reference shape by itself -- the GKR layer
    // prover sums its c variables out first, which leaves terms of this form over the b variables (gkr.hip)
    const uint64_t* lin_in;
    uint64_t* lin_out;
};
__device__ __forceinline__ Fr lds_load_fr(const uint32_t* base, uint32_t idx) {
    const uint4* p = reinterpret_cast<const uint4*>(base + 8 * idx);
    uint4 a = p[0], b = p[1];
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_store_fr(uint32_t* base, uint32_t idx, const Fr& v) {
    uint4* p = reinterpret_cast<uint4*>(base + 8 * idx);
    p[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
// evaluations at t = 0..K of lo + t (hi - lo), added into sums[0..K]
template <int K>
__device__ __forceinline__ void accumulate_linear_evals(const Fr& lo, const Fr& hi, Fr (&sums)[K + 1]) {
    Fr v = lo;
    const Fr d = hi - lo;
#pragma unroll
    for (int t = 0; t <= K; ++t) {
        sums[t] = sums[t] + v;
        if (t < K) v = v + d;
    }
}

// evaluations at t = 0..K of prod_k (lo_k + t*(hi_k - lo_k)), added into sums[0..K]
// K <= 2: a table's value at integer t by repeated addition of d = hi - lo, K - 1 products per t.
// K >= 3: the tables are taken in PAIRS.  A(t) = (lo_0 + t d_0)(lo_1 + t d_1) is a quadratic: A(0) = lo_0 lo_1, A(1) = hi_0 hi_1 and
// its leading coefficient d_0 d_1 -- three products -- give every A(t) by second differences (A(t+1) - A(t) grows by 2 d_0 d_1 per
// step: two additions per point).  K = 3: 3 + 4 products per index instead of 8; K = 4: 3 + 3 + 5 = 11 instead of 15; K = 5:
// 3 + 3 + 6 + 6 = 18 instead of 24.  These rounds are bound by the issue of multiply-adds (one product ~ 330 VALU instructions),
// not by memory: fewer products is the only lever.  Exact field arithmetic: the same canonical sums.
struct QuadEvals {          // a quadratic's values at t = 0, 1, 2, ...: cur = A(t), diff = A(t + 1) - A(t), dd = the second difference
    Fr cur, diff, dd;
    __device__ __forceinline__ QuadEvals(const Fr& lo0, const Fr& hi0, const Fr& lo1, const Fr& hi1) {
        const Fr a0 = lo0 * lo1, a1 = hi0 * hi1, lead = (hi0 - lo0) * (hi1 - lo1);
        cur = a0;
        diff = a1 - a0;
        dd = lead + lead;
    }
    __device__ __forceinline__ void step() { cur = cur + diff; diff = diff + dd; }
};
template <int K>
__device__ __forceinline__ void accumulate_round_evals(const Fr (&lo)[K], const Fr (&hi)[K], Fr (&sums)[K + 1]) {
    if constexpr (K <= 2) {
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
    } else if constexpr (K == 3) {
        QuadEvals a(lo[0], hi[0], lo[1], hi[1]);
        Fr v = lo[2];
        const Fr d = hi[2] - lo[2];
#pragma unroll
        for (int t = 0; t <= K; ++t) {
            sums[t] = sums[t] + a.cur * v;
            if (t < K) { a.step(); v = v + d; }
        }
    } else if constexpr (K == 4) {
        QuadEvals a(lo[0], hi[0], lo[1], hi[1]), b(lo[2], hi[2], lo[3], hi[3]);
#pragma unroll
        for (int t = 0; t <= K; ++t) {
            sums[t] = sums[t] + a.cur * b.cur;
            if (t < K) { a.step(); b.step(); }
        }
    } else {
        static_assert(K == 5, "product terms have at most five tables");
        QuadEvals a(lo[0], hi[0], lo[1], hi[1]), b(lo[2], hi[2], lo[3], hi[3]);
        Fr v = lo[4];
        const Fr d = hi[4] - lo[4];
#pragma unroll
        for (int t = 0; t <= K; ++t) {
            sums[t] = sums[t] + (a.cur * b.cur) * v;
            if (t < K) { a.step(); b.step(); v = v + d; }
        }
    }
}

// ComposedMultilinearTrait::element_wise_product / element_wise_add as materialised vectors
// (polynomial/src/composed/composed_multilinear.rs:105-119): out[i] = prod_k / sum_k table_k[i].  Up to 8 tables per launch;
// ACC continues a longer product from `out` itself.
struct ElementwisePtrs { const uint64_t* in[8]; };
template <bool PRODUCT>
static __global__ __launch_bounds__(MLE_BLOCK) void composed_elementwise_kernel(ElementwisePtrs t, uint32_t k, size_t n, uint32_t acc,
                                                                          uint64_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    for (size_t i = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; i < n; i += stride) {
        Fr v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) if ((uint32_t)q < k) v[q] = load_fr(t.in[q], i);
        Fr r = acc ? load_fr(out, i) : v[0];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if ((uint32_t)q < k && (acc || q > 0)) r = PRODUCT ? r * v[q] : r + v[q];
        }
        store_fr(out, i, r);
    }
}

// x / 2 mod r on the limbs (any representation: halving is linear): x even -> x >> 1, else (x + r) >> 1 (x + r < 2^256)
__device__ __forceinline__ Fr fr_half(const Fr& x) {
    uint32_t t[9];
    const uint32_t odd = x.l[0] & 1u;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t s = (uint64_t)x.l[i] + (odd ? FrParams::p(i) : 0u) + carry;
        t[i] = (uint32_t)s;
        carry = s >> 32;
    }
    t[8] = (uint32_t)carry;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = (t[i] >> 1) | (t[i + 1] << 31);
    return r;
}

// ---- closing a round ------------------------------------------------------------------------------------
struct ComposedMeta {
    uint32_t n_terms;
    uint32_t k[CMP_MAX_TERMS];         // tables per term (= degree of the term's round polynomial)
    uint32_t rec_off[CMP_MAX_TERMS];   // offset of the term's K+1 sums inside a workgroup record
    uint32_t rec;                      // sums per record
    uint32_t lin_tab[CMP_MAX_TERMS];   // tail kernel: index of the term's additive table among the LDS tables, or ~0u
    uint32_t multi;                    // 0: ComposedSumcheck transcript (raw evaluations), 1: multi-composed (sparse coefficients)
};
// Device-resident state.  interp[d] is the (d+1)x(d+1) matrix taking evaluations at x = 0..d to coefficients
// (Montgomery form), uploaded by the host once per prove.
struct ComposedDev {                   // one per context, persistent: a continuation finds the transcript where the last call left it
    Sha256State transcript;
    uint64_t last_canon[2][4];         // the challenge of round r as the hash yielded it (canonical), in slot r & 1: what the pipelined rounds evaluate
                                       // their forms at and fold by (composed_pipe.hpp).  Two slots: the workgroup that closes round r writes slot r & 1
                                       // while the others of the same launch still read round r - 1's
    uint64_t interp[CMP_MAX_K + 1][(CMP_MAX_K + 1) * (CMP_MAX_K + 1)][4];
};
struct CloseShared {                   // LDS scratch of close_round
    Fr evals[CMP_MAX_REC];             // the round's sums: term p's evaluations at t = 0..K_p from rec_off[p]
    Fr prod[CMP_MAX_REC * (CMP_MAX_K + 1)];   // interpolation products, (record entry, i)
    Fr interp_c[CMP_MAX_TERMS * (CMP_MAX_K + 1) * (CMP_MAX_K + 1)];   // this lane's interpolation matrix entry (close_preload), by product index
    Fr term_coeff[CMP_MAX_TERMS][CMP_MAX_MONO];
    Fr canon[CMP_MAX_MONO];            // canonical integers of what the transcript absorbs, in order
    uint32_t pow_of[CMP_MAX_MONO];
    uint32_t n_items;
    Fr challenge_canon;                // the round's challenge as the hash yields it (canonical integer), for the caller
    Fr sum_canon;                      // canonical claimed sum (absorbed in the first round of a multi-composed proof)
    uint32_t msg[16 * CMP_MAX_BLOCKS]; // the padded message of the round: pending bytes || items || padding
    uint32_t kw[64 * CMP_MAX_BLOCKS];  // K + W of every block
    uint32_t kw_ready[CMP_MAX_BLOCKS]; // chunks of 16 words of kw published so far, per block (sha256_schedule_to_lds)
    uint32_t n_blocks;
};

// ---- an OUTER transcript fed beside the rounds ---------------------------------------------------------------------------------
// GKRProtocol::prove absorbs every layer's ComposedSumcheckProof::to_bytes() into its own transcript (gkr/src/protocol.rs:91,
// gkr/src/utils.rs:38) and draws alpha, beta from it (:104-105): the same item bytes the rounds' own transcript absorbs, in a second
// hash chain.  On the host that was one synchronisation per layer.  Here every closing kernel is launched with ONE MORE workgroup, the
// hasher: the wave that closes a round publishes its items (canonical coefficient, power) in a ring in global memory and raises the
// round's flag (release, agent scope, by a wave that is not on the round's critical path); the hasher waits for the flag (acquire),
// absorbs the items into the outer state -- which lives in global memory between kernels, like the rounds' own transcript -- and ends
// with the kernel.  Its ~2 us per 64-byte item run beside the rounds' own hash of the same bytes.
static_assert(CMP_OUTER_MONO == CMP_MAX_MONO, "the ring holds a whole round polynomial");
struct CloseArgs {
    ComposedMeta meta;
    ComposedDev* st;
    FrArg sum;                          // the claimed sum (absorbed in the first round of a multi-composed proof) ...
    const uint64_t* sum_dev;            // ... or, when not null, where it lies in device memory (the kernel before computed it)
    uint32_t round, first;
    uint64_t* round_out;
    uint64_t* challenges;
    OuterPub outer;
};
__device__ __forceinline__ Fr close_claimed_sum(const CloseArgs& ca) { return ca.sum_dev ? load_fr(ca.sum_dev, 0) : fr_from_arg(ca.sum); }
// one whole wave, after the round's items are in sh (canon, pow_of, n_items) and a barrier has made them visible to it.
// One row = 64 words = one agent-scope store per lane, then the flag behind a release fence at agent scope (the reader may run on
// another XCD, behind another L2).
__device__ __forceinline__ void outer_publish(const OuterPub& op, const CloseShared& sh, uint32_t round) {
    if (!op.dev) return;
    const uint32_t lane = threadIdx.x & 63, n = sh.n_items;
    uint32_t* it = op.dev->items[round];
    uint32_t v = n;
    if (lane >= 1) {
        const uint32_t w = lane - 1, i = w / 9, o = w - 9 * i;
        v = i < n ? (o < 8 ? sh.canon[i].l[o] : sh.pow_of[i]) : 0u;
    }
    __hip_atomic_store(&it[lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // release at agent scope: the row before the flag, for a reader on another XCD.  (The write-back this implies runs on a wave that
    // has nothing else to do while wave 0 hashes; measured: no difference to a relaxed flag -- which two boxes of four processes
    // sharing one GPU showed to be not enough.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(&op.dev->flag[round], op.token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The hasher workgroup (every thread of it calls; >= 128 threads): absorbs rounds [first_round, first_round + n_rounds) as they are
// published.  transcript.commit(&sumcheck_proof.to_bytes()): per round, per monomial coeff || pow as 32-byte big-endian integers
// (multi_composed_sumcheck.rs:24-31, sparse_univariate.rs:27-34) -- 64 bytes per monomial, so a round of n monomials is exactly n
// blocks whatever is pending (the pending part is 0 or 32 bytes and stays so).  Like the rounds' own hash: the waves beside wave 0
// compute the message schedules of all blocks side by side, wave 0 runs the state rounds only (~1.7 us per block).
struct OuterShared {
    uint32_t row[64];
    uint32_t msg[16 * CMP_OUTER_MONO];
    uint32_t kw[64 * CMP_OUTER_MONO];
    uint32_t buf[8];
    uint32_t fill_words, n, ok;
};
__device__ __forceinline__ uint32_t outer_stream_word(const uint32_t* row, uint32_t x) {      // word x of the round's item bytes
    const uint32_t i = x >> 4, o = x & 15;
    return o < 8 ? row[1 + 9 * i + (7 - o)] : (o == 15 ? row[1 + 9 * i + 8] : 0u);
}
__device__ __noinline__ void outer_absorb_rounds(OuterPub op, uint32_t first_round, uint32_t n_rounds) {
    __shared__ OuterShared os;
    static_assert(1 + CMP_OUTER_MONO * 9 == 64, "a round's row is one word per lane");
    const uint32_t tid = threadIdx.x, wave = tid >> 6, n_sched = (blockDim.x >> 6) - 1;
    uint32_t h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = op.dev->state.h[i];
    uint64_t len = op.dev->state.len;
    if (tid < 8) os.buf[tid] = op.dev->state.buf[tid];
    if (tid == 0) os.fill_words = op.dev->state.fill >> 2;
    __syncthreads();
    for (uint32_t r = first_round; r < first_round + n_rounds; ++r) {
        if (wave == 0) {
            uint32_t spins = 0;
            bool ok = true;
            while (__hip_atomic_load(&op.dev->flag[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != op.token) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 25)) { ok = false; break; }     // seconds: something upstream failed; say so instead of hanging the stream
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the row after the flag
            const uint32_t mine = ok ? __hip_atomic_load(&op.dev->items[r][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            os.row[tid] = mine;
            if (tid == 0) { os.ok = ok ? 1u : 0u; os.n = mine < (uint32_t)CMP_OUTER_MONO ? mine : (uint32_t)CMP_OUTER_MONO; }
        }
        __syncthreads();
        if (!os.ok) {
            if (tid == 0) op.dev->error = 1 + r;
            break;
        }
        const uint32_t n = os.n, fill = os.fill_words;
        for (uint32_t t = tid; t < 16 * n; t += blockDim.x) {
            const uint32_t b = t >> 4, k = t & 15;
            uint32_t v;
            if (fill == 0) v = outer_stream_word(os.row, t);
            else v = k < 8 ? (b == 0 ? os.buf[k] : outer_stream_word(os.row, 16 * (b - 1) + 8 + k)) : outer_stream_word(os.row, 16 * b + (k - 8));
            os.msg[t] = v;
        }
        __syncthreads();
        if (wave >= 1) {
            for (uint32_t b0 = 4 * (wave - 1); b0 < n; b0 += 4 * n_sched) {
                const uint32_t b = b0 + ((tid >> 4) & 3);
                const bool active = b < n;
                const uint32_t bb = active ? b : b0;
                sha256_schedule_rows_to_lds(os.msg[16 * bb + (tid & 15)], os.kw + 64 * bb, nullptr, 0u, 0u, active);
            }
        }
        __syncthreads();
        if (wave == 0) {
            ShaSplit sp;
            sp.init();
            uint32_t hs[4];
            sp.split(h, hs);
            for (uint32_t b = 0; b < n; ++b) sha256_rounds_block_split(sp, hs, os.kw + 64 * b);
            sp.join(hs, h);
            if (fill != 0 && n > 0 && tid < 8) os.buf[tid] = outer_stream_word(os.row, 16 * (n - 1) + 8 + tid);
        }
        len += 64ull * n;
        __syncthreads();
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) op.dev->state.h[i] = h[i];
        op.dev->state.len = len;
    }
    if (tid < 8) op.dev->state.buf[tid] = os.buf[tid];      // (fill is unchanged: whole 64-byte items only)
}

// Closes a round with the whole workgroup (>= 64 threads; every thread must call): sh.evals hold the sums.
// Builds the round polynomial, absorbs it, derives the challenge (left in sh.challenge_canon and, in Montgomery form, challenges[round]).
//   first: 1 = the transcript is started here (ComposedSumcheck: nothing absorbed before, composed_sumcheck.rs:33;
//              multi-composed prove_partial: the claimed sum, multi_composed_sumcheck.rs:60,70),
//          2 = the transcript state was prepared by the host (multi-composed `prove`: all table bytes were
//              hashed first, :51-53) and the claimed sum is absorbed here,
//          0 = continue.
// Output per round (round_out + 64 * round, in u64):
//   multi == 0: K+1 evaluations (4 u64 each);  multi == 1: [0] = #monomials, then (coeff, pow) pairs of 8 u64 from [8].
// Only the hash chain is serial (thread 0); interpolation products and Montgomery conversions run one per lane.
// Once per kernel, before the first close_round (every thread; a barrier must follow before the round is closed): the
// interpolation matrix entries the lanes multiply by, out of global memory into LDS -- a 2 us load on every round's critical
// path otherwise.
__device__ __forceinline__ void close_preload(CloseShared& sh, const ComposedMeta& meta, const ComposedDev* st) {
    if (!meta.multi) return;
    const uint32_t tid = threadIdx.x;
    uint32_t p = 0, base = 0;
    while (p < meta.n_terms && tid >= base + (meta.k[p] + 1) * (meta.k[p] + 1)) { base += (meta.k[p] + 1) * (meta.k[p] + 1); ++p; }
    if (p < meta.n_terms) {
        const uint32_t d = meta.k[p], e = tid - base;
        sh.interp_c[tid] = load_fr(&st->interp[d][e][0], 0);
    }
}
// shadow(lane, n_lanes): optional work of the caller's that needs neither the round's challenge nor the hash wave; the waves that
// compute the message schedules run it once they are done, i.e. beside the hash (lane < n_lanes = blockDim.x - 64).
// On return sh.challenge_canon holds the challenge as a CANONICAL integer (what the hash yields) for every thread; its Montgomery
// form goes to challenges[round] without anyone waiting for the conversion.
struct NoShadow { __device__ __forceinline__ void operator()(uint32_t, uint32_t) const {} };
template <class Shadow = NoShadow>
__device__ __forceinline__ void close_round(CloseShared& sh, const CloseArgs& ca, Sha256State* tr_state,
                                            uint32_t round, uint32_t first, const Shadow& shadow = Shadow()) {
    const ComposedMeta& meta = ca.meta;
    ComposedDev* st = ca.st;
    uint64_t* __restrict__ challenges = ca.challenges;
    uint64_t* out = ca.round_out + 64 * (size_t)round;
    const uint32_t tid = threadIdx.x;
    ZK_STAMP_AT(0, round, 0);
    if (!meta.multi) {
        // transcript.commit(&vec_to_bytes(&round_poly))  composed_sumcheck.rs:51: the raw evaluations
        if (tid <= meta.k[0]) {
            const Fr e = sh.evals[tid];
            store_fr(out, tid, e);
            sh.canon[tid] = fr_from_mont_outlined(e);
        }
        if (tid == 0) sh.n_items = meta.k[0] + 1;
    } else {
        // round_poly = sum over terms of interpolation(evals at x = 0..K)  (:79-95); coefficients that are zero
        // are dropped per term (sparse_univariate.rs:55) but a zero produced by the sum is kept (:159-203).
        // Degrees <= 2 (every GKR term) need no interpolation matrix: c0 = e0, c2 = (e0 - 2 e1 + e2) / 2, c1 = e1 - e0 - c2 -- additions
        // and one halving per lane instead of a product, a sum of products and a barrier between them (2 us of every round).
        // Higher degrees: coefficient k of term p = sum_i interp[d][k][i] * eval_i, one lane per product (p, k, i) -- at most
        // 4 * 36 of them -- then one lane per coefficient adds its d + 1 products.
        bool low = true;
        for (uint32_t p = 0; p < meta.n_terms; ++p) low = low && meta.k[p] <= 2;
        if (low) {
            if (tid < meta.rec) {
                uint32_t p = 0;
                while (p + 1 < meta.n_terms && tid >= meta.rec_off[p + 1]) ++p;
                const uint32_t d = meta.k[p], k = tid - meta.rec_off[p];
                const Fr e0 = sh.evals[meta.rec_off[p]], e1 = sh.evals[meta.rec_off[p] + 1];
                Fr c = e0;
                if (k > 0) {
                    if (d == 1) c = e1 - e0;
                    else {
                        const Fr c2 = fr_half(e0 - (e1 + e1) + sh.evals[meta.rec_off[p] + 2]);
                        c = k == 2 ? c2 : e1 - e0 - c2;
                    }
                }
                sh.term_coeff[p][k] = c;
            }
        } else {
            {
                uint32_t p = 0, base = 0;
                while (p < meta.n_terms && tid >= base + (meta.k[p] + 1) * (meta.k[p] + 1)) { base += (meta.k[p] + 1) * (meta.k[p] + 1); ++p; }
                if (p < meta.n_terms) {
                    const uint32_t d = meta.k[p], e = tid - base, k = e / (d + 1), i = e % (d + 1);
                    sh.prod[(meta.rec_off[p] + k) * (CMP_MAX_K + 1) + i] = fr_mul_outlined(sh.interp_c[tid], sh.evals[meta.rec_off[p] + i]);
                }
            }
            __syncthreads();
            if (tid < meta.rec) {
                uint32_t p = 0;
                while (p + 1 < meta.n_terms && tid >= meta.rec_off[p + 1]) ++p;
                const uint32_t d = meta.k[p], k = tid - meta.rec_off[p];
                Fr c = sh.prod[tid * (CMP_MAX_K + 1)];
                for (uint32_t i = 1; i <= d; ++i) c = c + sh.prod[tid * (CMP_MAX_K + 1) + i];
                sh.term_coeff[p][k] = c;
            }
        }
        __syncthreads();
        if (tid < 64) {   // lanes 0..6 of wave 0: one power each
            Fr coeff = Fr::zero();
            bool present = false;
            if (tid < CMP_MAX_MONO) {
                for (uint32_t p = 0; p < meta.n_terms; ++p) {
                    if (tid > meta.k[p]) continue;
                    const Fr c = sh.term_coeff[p][tid];
                    if (!c.is_zero()) { coeff = coeff + c; present = true; }
                }
            }
            const uint64_t mask = __ballot(present);
            if (present) {
                const uint32_t at = __popcll(mask & (((uint64_t)1 << tid) - 1));
                sh.canon[at] = fr_from_mont_outlined(coeff);
                sh.pow_of[at] = tid;
                store_fr(out + 8, 2 * at, coeff);
                Fr powm = Fr::zero();                        // Fr::from(tid): tid <= 5 additions of one instead of a product
                for (uint32_t i = 0; i < tid; ++i) powm = powm + Fr::one();
                store_fr(out + 8, 2 * at + 1, powm);
            }
            if (tid == 0) { sh.n_items = (uint32_t)__popcll(mask); out[0] = (uint64_t)__popcll(mask); }
        }
    }
    if (meta.multi && first && tid == 64) sh.sum_canon = fr_from_mont_outlined(close_claimed_sum(ca));   // multi_composed_sumcheck.rs:70
    __syncthreads();
    ZK_STAMP_AT(0, round, 1);
    // ---- the round's message, padded (FiatShamirTranscript: commit ... then challenge = finalize, fiat_shamir.rs:17-25):
    // what the hasher still holds || [claimed sum] || items || 0x80 00.. || bit length.  Every word is written by one lane.
    {
        const uint32_t pending = (first == 1) ? 0u : (tr_state->fill >> 2);        // words
        const uint64_t len_prev = (first == 1) ? 0u : tr_state->len;               // bytes, pending included
        const uint32_t sum_words = (meta.multi && first) ? 8u : 0u;
        const uint32_t item_words = meta.multi ? 16u : 8u;
        const uint32_t n_items = sh.n_items;
        const uint32_t body = pending + sum_words + n_items * item_words;
        const uint32_t n_blocks = (body + 3 + 15) / 16;                             // + 0x80 word + 64-bit length
        const uint64_t bits = (len_prev + 4ull * (sum_words + n_items * item_words)) * 8ull;
        for (uint32_t w = tid; w < 16 * n_blocks; w += blockDim.x) {
            uint32_t v = 0;
            if (w < pending) v = tr_state->buf[w];
            else if (w < pending + sum_words) v = sh.sum_canon.l[7 - (w - pending)];
            else if (w < body) {
                const uint32_t q = w - pending - sum_words, i = q / item_words, o = q % item_words;
                v = o < 8 ? sh.canon[i].l[7 - o] : (o == 15 ? sh.pow_of[i] : 0u);   // coeff || pow, big-endian (sparse_univariate.rs:27-34)
            } else if (w == body) v = 0x80000000u;
            else if (w == 16 * n_blocks - 2) v = (uint32_t)(bits >> 32);
            else if (w == 16 * n_blocks - 1) v = (uint32_t)bits;
            sh.msg[w] = v;
        }
        if (tid == 0) sh.n_blocks = n_blocks;
        if (tid < CMP_MAX_BLOCKS) sh.kw_ready[tid] = 0;
    }
    __syncthreads();
    ZK_STAMP_AT(0, round, 2);
    // Wave 0 runs the state rounds of every block while the other waves compute the message schedules: the first block's rounds
    // 0..15 come straight from the message and the rest from kw as it is published in chunks of 16 words (sha256_schedule_to_lds /
    // sha256_compress_kw), so the hash does not wait for whole schedules as it did (1.75 us per round).
    ZK_STAMP_AT(0, round, 3);
    if (tid >= 64) {
        const uint32_t wave = (tid >> 6) - 1, n_sched = (blockDim.x >> 6) - 1;
        for (uint32_t b0 = 4 * wave; b0 < sh.n_blocks; b0 += 4 * n_sched) {    // a block per row of 16 lanes (sha256_schedule_rows_to_lds)
            const uint32_t b = b0 + ((tid >> 4) & 3);
            const bool active = b < sh.n_blocks;
            const uint32_t bb = active ? b : b0;
            sha256_schedule_rows_to_lds(sh.msg[16 * bb + (tid & 15)], sh.kw + 64 * bb, &sh.kw_ready[bb], 0u, bb == 0 ? 1u : 0u, active);
        }
        if (wave == n_sched - 1) outer_publish(ca.outer, sh, round);      // the last wave: the one with the fewest schedules
        shadow(tid - 64, blockDim.x - 64);
    } else {
        uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        if (first != 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = tr_state->h[i];
        }
        const uint32_t n_blocks = sh.n_blocks;
        // block 0 chunk by chunk; the later blocks' schedules (1.8 us each, started together with block 0's) are complete when the
        // hash reaches them: one wait, then the plain rounds
        sha256_message_split(h, sh.msg, sh.kw, sh.kw_ready, n_blocks);
        ZK_STAMP_AT(0, round, 4);
        Fr c;                                                   // from_be_bytes_mod_order (fiat_shamir.rs:27-29)
#pragma unroll
        for (int i = 0; i < 8; ++i) c.l[i] = h[7 - i];
        c.reduce_once();
        c.reduce_once();
        if (tid == 0) {
            Transcript tr;                                      // finalize_reset + update(digest)
            tr.init();
            tr.commit_words8(h);
            tr.store(tr_state);
            sh.challenge_canon = c;
#pragma unroll
            for (int i = 0; i < Fr::N; ++i) reinterpret_cast<uint32_t*>(st->last_canon[round & 1])[i] = c.l[i];
        }
    }
    __syncthreads();
    if (tid == 0) store_fr(challenges, round, fr_to_mont_outlined(sh.challenge_canon));   // nobody in this kernel waits for it
    ZK_STAMP_AT(0, round, 5);
}

// One round of one product term.
//   FOLD = false: evaluate the round polynomial of the tables as they are (first round): pairs (j, j + n/2).
//   FOLD = true : fold every table at *r_ptr first (in[j], in[j + n/2] -> out[j]) for the two outputs j and
//                 j + n/4, store them, and evaluate the NEXT round polynomial on that folded pair.
// Per workgroup, the K+1 sums go to partials[(block * rec + rec_off + t)]; composed_close_kernel sums the records
// and closes the round.  (Closing inside this kernel by the last workgroup to finish was measured and dropped: on
// eight XCDs with private L2s the agent-scope release every workgroup then needs costs more than a launch.)
// (lo, hi) of table k at output pair j: the entries themselves (first round) or the fold at r of the four entries behind them,
// written back on the way
template <bool FOLD>
__device__ __forceinline__ void round_pair(const TablePtrs& tp, int k, size_t j, size_t h, size_t q, const Fr& r, Fr& lo, Fr& hi) {
    if constexpr (FOLD) {
        const Fr a0 = load_fr(tp.in[k], j), b0 = load_fr(tp.in[k], j + h), a1 = load_fr(tp.in[k], j + q), b1 = load_fr(tp.in[k], j + h + q);
        lo = fold_pair(a0, b0, r);
        hi = fold_pair(a1, b1, r);
        store_fr(tp.out[k], j, lo);
        store_fr(tp.out[k], j + q, hi);
    } else {
        lo = load_fr(tp.in[k], j);
        hi = load_fr(tp.in[k], j + h);
    }
}
// accumulate_round_evals for K >= 3 with the tables taken from memory two at a time: a pair of tables is consumed into its quadratic
// before the next pair is loaded (all K tables in registers at once is what held these kernels at one wave per SIMD)
template <int K, bool FOLD>
__device__ __forceinline__ void accumulate_round_evals_streamed(const TablePtrs& tp, size_t j, size_t h, size_t q, const Fr& r, Fr (&sums)[K + 1]) {
    static_assert(K >= 3 && K <= 5, "pairs of tables");
    Fr l0, h0, l1, h1;
    round_pair<FOLD>(tp, 0, j, h, q, r, l0, h0);
    round_pair<FOLD>(tp, 1, j, h, q, r, l1, h1);
    QuadEvals a(l0, h0, l1, h1);
    if constexpr (K == 3) {
        round_pair<FOLD>(tp, 2, j, h, q, r, l0, h0);
        Fr v = l0;
        const Fr d = h0 - l0;
#pragma unroll
        for (int t = 0; t <= K; ++t) {
            sums[t] = sums[t] + a.cur * v;
            if (t < K) { a.step(); v = v + d; }
        }
    } else {
        round_pair<FOLD>(tp, 2, j, h, q, r, l0, h0);
        round_pair<FOLD>(tp, 3, j, h, q, r, l1, h1);
        QuadEvals b(l0, h0, l1, h1);
        if constexpr (K == 4) {
#pragma unroll
            for (int t = 0; t <= K; ++t) {
                sums[t] = sums[t] + a.cur * b.cur;
                if (t < K) { a.step(); b.step(); }
            }
        } else {
            round_pair<FOLD>(tp, 4, j, h, q, r, l0, h0);
            Fr v = l0;
            const Fr d = h0 - l0;
#pragma unroll
            for (int t = 0; t <= K; ++t) {
                sums[t] = sums[t] + (a.cur * b.cur) * v;
                if (t < K) { a.step(); b.step(); v = v + d; }
            }
        }
    }
}
template <int K, bool FOLD, bool LIN>
static __global__ __launch_bounds__(MLE_BLOCK) __attribute__((amdgpu_waves_per_eu(K >= 4 ? 2 : 1))) void composed_round_kernel(TablePtrs tp, size_t n, const uint64_t* __restrict__ r_ptr,
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
                if (LIN) store_fr(tp.lin_out, 0, fold_pair(load_fr(tp.lin_in, 0), load_fr(tp.lin_in, 1), r));
            }
        }
        for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < q; j += stride) {
            if constexpr (K >= 3 && !LIN) {
                accumulate_round_evals_streamed<K, true>(tp, j, h, q, r, sums);
            } else {
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
                if (LIN) {
                    const Fr llo = fold_pair(load_fr(tp.lin_in, j), load_fr(tp.lin_in, j + h), r);
                    const Fr lhi = fold_pair(load_fr(tp.lin_in, j + q), load_fr(tp.lin_in, j + h + q), r);
                    store_fr(tp.lin_out, j, llo);
                    store_fr(tp.lin_out, j + q, lhi);
                    accumulate_linear_evals<K>(llo, lhi, sums);
                }
            }
        }
    } else {
        const size_t h = n >> 1;
        for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < h; j += stride) {
            if constexpr (K >= 3 && !LIN) {
                accumulate_round_evals_streamed<K, false>(tp, j, h, 0, Fr::zero(), sums);
            } else {
                Fr lo[K], hi[K];
#pragma unroll
                for (int k = 0; k < K; ++k) { lo[k] = load_fr(tp.in[k], j); hi[k] = load_fr(tp.in[k], j + h); }
                accumulate_round_evals<K>(lo, hi, sums);
                if (LIN) accumulate_linear_evals<K>(load_fr(tp.lin_in, j), load_fr(tp.lin_in, j + h), sums);
            }
        }
    }
#pragma unroll
    for (int t = 0; t <= K; ++t) {
        Fr s = block_reduce_fr(sums[t], red);
        if (threadIdx.x == 0) store_fr(partials, (size_t)blockIdx.x * rec + rec_off + t, s);
    }
}

// The folding round for SMALL tables of claims with a term of K >= 3 tables (<= CMP_TSPLIT_MAX output pairs): K + 1 LANES per output
// pair, every term of the claim in ONE launch (blockIdx.y = term; K is a run-time value per term).
// With one lane per pair such a round is 2 K + (K + 1)(K - 1) Montgomery products one after another on every lane -- 34 at K = 5,
// ~26 us on a lone wave, whatever the size of the tables -- and a chip with 1024 SIMDs has nothing else to do.  Here lane t of a
// group first folds table t (lanes 0..K-1: two products), the group exchanges the folded values through LDS, and then lane t
// evaluates the product at ITS point t: K - 1 products.  6 products deep instead of 34; a third more work in total, so only below
// the size where one lane per pair is latency.  Same values, same record layout as composed_round_kernel.
constexpr size_t CMP_TSPLIT_MAX = 65536;
struct TsplitTerms {
    TablePtrs t[CMP_MAX_TERMS];
    uint32_t rec_off[CMP_MAX_TERMS];
    uint32_t k[CMP_MAX_TERMS];
};
static __global__ __launch_bounds__(MLE_BLOCK) void composed_round_tsplit_kernel(TsplitTerms mp, size_t n, const uint64_t* __restrict__ r_ptr,
                                                                          uint32_t rec, uint64_t* __restrict__ partials) {
    constexpr int NW = MLE_BLOCK / 64;
    constexpr int XCH = 104;                                             // (64 / (K + 1)) K 2 field elements per wave, largest at K = 5 (100)
    __shared__ __attribute__((aligned(16))) uint32_t xch[NW * XCH * 8];   // folded (lo, hi) of every table of every group
    __shared__ __attribute__((aligned(16))) uint32_t part[NW * 64 * 8];   // every lane's sum
    const TablePtrs& tp = mp.t[blockIdx.y];
    const uint32_t K = mp.k[blockIdx.y], rec_off = mp.rec_off[blockIdx.y];
    const uint32_t G = K + 1, GPW = 64 / G;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = lane / G, t = lane - g * G;
    const bool active = g < GPW;
    const size_t h = n >> 1, q = n >> 2;
    const Fr r = load_fr(r_ptr, 0);
    // this lane's table in the folding phase (lanes t < K)
    const uint64_t* in = tp.in[0];
    uint64_t* out = tp.out[0];
    for (uint32_t k = 1; k < K; ++k) if (t == k) { in = tp.in[k]; out = tp.out[k]; }
    if (q == 0 && blockIdx.x == 0 && lane < K && wave == 0)          // n == 2: one entry per table is left, no pair to sum over
        store_fr(out, 0, fold_pair(load_fr(in, 0), load_fr(in, 1), r));
    Fr sum = Fr::zero();
    const size_t per_wg = (size_t)NW * GPW;
    uint32_t* my = xch + ((size_t)wave * XCH + (size_t)(active ? g : 0) * K * 2) * 8;
    for (size_t base = (size_t)blockIdx.x * per_wg; base < q; base += (size_t)gridDim.x * per_wg) {    // uniform per workgroup
        const size_t j = base + (size_t)wave * GPW + g;
        const bool valid = active && j < q;
        if (valid && t < K) {                                           // lane t folds table t
            const Fr a0 = load_fr(in, j), b0 = load_fr(in, j + h), a1 = load_fr(in, j + q), b1 = load_fr(in, j + h + q);
            const Fr lo = fold_pair(a0, b0, r), hi = fold_pair(a1, b1, r);
            store_fr(out, j, lo);
            store_fr(out, j + q, hi);
            lds_store_fr(my, 2 * t, lo);
            lds_store_fr(my, 2 * t + 1, hi);
        }
        __syncthreads();
        if (valid) {                                                    // lane t evaluates prod_k (lo_k + t d_k)
            Fr prod = Fr::zero();
            for (uint32_t k = 0; k < K; ++k) {
                const Fr lo = lds_load_fr(my, 2 * k), hi = lds_load_fr(my, 2 * k + 1);
                // v = hi + (t - 1) d for t >= 1: the bits of t - 1 select d, 2 d, 4 d
                const Fr d = hi - lo, d2 = d + d, d4 = d2 + d2;
                const uint32_t e = t - 1;                               // t = 0: unused
                Fr v = hi;
                v = v + ((e & 1u) ? d : Fr::zero());
                v = v + ((e & 2u) ? d2 : Fr::zero());
                v = v + ((e & 4u) ? d4 : Fr::zero());
                if (t == 0) v = lo;
                prod = k == 0 ? v : prod * v;
            }
            sum = sum + prod;
        }
        __syncthreads();                                                // before the next pass overwrites the exchange area
    }
    lds_store_fr(part, threadIdx.x, active ? sum : Fr::zero());
    __syncthreads();
    if (threadIdx.x <= K) {                                             // thread t: the sum over every group of every wave
        Fr s = Fr::zero();
        for (uint32_t w = 0; w < (uint32_t)NW; ++w)
            for (uint32_t gg = 0; gg < GPW; ++gg) s = s + lds_load_fr(part, w * 64 + gg * G + threadIdx.x);
        store_fr(partials, (size_t)blockIdx.x * rec + rec_off + threadIdx.x, s);
    }
}

// The K = 2 round of ONE term without additive table on LARGE tables (>= CMP_WIDE_MIN_WORK pairs): the round polynomial
// p(t) = sum_j (lo0 + t d0)(lo1 + t d1) is fixed by E0 = sum lo0 lo1, E1 = sum hi0 hi1 and D = sum d0 d1 (p(2) = 2 E1 - E0 + 2 D),
// and the three sums of products are accumulated UNREDUCED (wide_acc.hpp): a pair costs 3 x (64 mads + carries) instead of 3
// Montgomery products + 9 modular additions.  The three 9-word reductions + rescalings at the end cost ~2.5 pairs' worth, so a
// lane takes >= 8-16 pairs (a small grid; the three accumulators also hold the kernel at two waves per SIMD) -- which is why
// only the first rounds of a large claim come here.  Exact arithmetic: the same canonical values as composed_round_kernel.
constexpr size_t CMP_WIDE_MIN_WORK = (size_t)1 << 20;
template <bool FOLD>
static __global__ __launch_bounds__(MLE_BLOCK) void composed_round_wide2_kernel(TablePtrs tp, size_t n, const uint64_t* __restrict__ r_ptr,
                                                                         uint32_t rec, uint32_t rec_off, uint64_t* __restrict__ partials) {
    __shared__ Fr red[MLE_BLOCK / 64];
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    WideAcc e0, e1, dd;
    e0.clear(); e1.clear(); dd.clear();
    if (FOLD) {
        const Fr r = load_fr(r_ptr, 0);
        const size_t h = n >> 1, q = n >> 2;
        for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < q; j += stride) {
            Fr lo[2], hi[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const Fr a0 = load_fr(tp.in[k], j), a1 = load_fr(tp.in[k], j + q);
                const Fr b0 = load_fr(tp.in[k], j + h), b1 = load_fr(tp.in[k], j + h + q);
                lo[k] = fold_pair(a0, b0, r);
                hi[k] = fold_pair(a1, b1, r);
                store_fr(tp.out[k], j, lo[k]);
                store_fr(tp.out[k], j + q, hi[k]);
            }
            e0.mac(lo[0], lo[1]);
            e1.mac(hi[0], hi[1]);
            dd.mac(hi[0] - lo[0], hi[1] - lo[1]);
        }
    } else {
        const size_t h = n >> 1;
        for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < h; j += stride) {
            const Fr lo0 = load_fr(tp.in[0], j), hi0 = load_fr(tp.in[0], j + h);
            const Fr lo1 = load_fr(tp.in[1], j), hi1 = load_fr(tp.in[1], j + h);
            e0.mac(lo0, lo1);
            e1.mac(hi0, hi1);
            dd.mac(hi0 - lo0, hi1 - lo1);
        }
    }
    const Fr fix = fr_mont_2_32();
    Fr sums[3];
    sums[0] = wide_reduce(e0.lo, e0.hi) * fix;
    sums[1] = wide_reduce(e1.lo, e1.hi) * fix;
    const Fr D = wide_reduce(dd.lo, dd.hi) * fix;
    sums[2] = (sums[1] + sums[1] - sums[0]) + (D + D);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const Fr s = block_reduce_fr(sums[t], red);
        if (threadIdx.x == 0) store_fr(partials, (size_t)blockIdx.x * rec + rec_off + t, s);
    }
}

// The same round for ALL terms of a multi-composed claim whose terms have the same number of tables (GKR: two terms of two
// tables, one of them with an additive table): blockIdx.y is the term, so a round is one launch instead of one per term --
// the rounds of a layer proof are launch-latency sized (13 us per launch at 2^20 entries and below).
struct MultiTablePtrs {
    TablePtrs t[CMP_MAX_TERMS];
    uint32_t rec_off[CMP_MAX_TERMS];
};
template <int K, bool FOLD>
static __global__ __launch_bounds__(MLE_BLOCK) void composed_round_multi_kernel(MultiTablePtrs mp, size_t n, const uint64_t* __restrict__ r_ptr,
                                                                          uint32_t rec, uint64_t* __restrict__ partials) {
    __shared__ Fr red[MLE_BLOCK / 64];
    const TablePtrs& tp = mp.t[blockIdx.y];
    const uint32_t rec_off = mp.rec_off[blockIdx.y];
    const bool lin = tp.lin_in != nullptr;
    const size_t stride = (size_t)gridDim.x * MLE_BLOCK;
    Fr sums[K + 1];
#pragma unroll
    for (int t = 0; t <= K; ++t) sums[t] = Fr::zero();
    if (FOLD) {
        const Fr r = load_fr(r_ptr, 0);
        const size_t h = n >> 1, q = n >> 2;
        if (q == 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) store_fr(tp.out[k], 0, fold_pair(load_fr(tp.in[k], 0), load_fr(tp.in[k], 1), r));
                if (lin) store_fr(tp.lin_out, 0, fold_pair(load_fr(tp.lin_in, 0), load_fr(tp.lin_in, 1), r));
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
            if (lin) {
                const Fr llo = fold_pair(load_fr(tp.lin_in, j), load_fr(tp.lin_in, j + h), r);
                const Fr lhi = fold_pair(load_fr(tp.lin_in, j + q), load_fr(tp.lin_in, j + h + q), r);
                store_fr(tp.lin_out, j, llo);
                store_fr(tp.lin_out, j + q, lhi);
                accumulate_linear_evals<K>(llo, lhi, sums);
            }
        }
    } else {
        const size_t h = n >> 1;
        for (size_t j = (size_t)blockIdx.x * MLE_BLOCK + threadIdx.x; j < h; j += stride) {
            Fr lo[K], hi[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { lo[k] = load_fr(tp.in[k], j); hi[k] = load_fr(tp.in[k], j + h); }
            accumulate_round_evals<K>(lo, hi, sums);
            if (lin) accumulate_linear_evals<K>(load_fr(tp.lin_in, j), load_fr(tp.lin_in, j + h), sums);
        }
    }
#pragma unroll
    for (int t = 0; t <= K; ++t) {
        Fr s = block_reduce_fr(sums[t], red);
        if (threadIdx.x == 0) store_fr(partials, (size_t)blockIdx.x * rec + rec_off + t, s);
    }
}

// The same folding round for SMALL tables of K = 2 terms (<= CMP_SPLIT_MAX output pairs): FOUR lanes per output pair.  With one
// lane per pair a round is 9-11 Montgomery products one after another on a lone wave (0.9 us each: the chip has far more SIMDs
// than such a round has waves), which is what made a small round kernel 13 us; here lane g of a group folds ONE of the four
// values (table g / 2, lower or upper output), the group exchanges them, and lanes 0 / 1 / 2 multiply for t = 0 / 1 / 2: two
// products deep (three with an additive table).  Same values, same record layout (gridDim.x records).
constexpr size_t CMP_SPLIT_MAX = 65536;   // measured: 16384 / 65536 / 262144 output pairs -> 0.710 / 0.705 / 0.72 ms (K = 2, 2^22), 11.63 / 11.57 / 11.70 ms (GKR depth 20)
static __global__ __launch_bounds__(MLE_BLOCK) void composed_round_split2_kernel(MultiTablePtrs mp, size_t n, const uint64_t* __restrict__ r_ptr,
                                                                          uint32_t rec, uint64_t* __restrict__ partials) {
    __shared__ Fr red[3][MLE_BLOCK / 64];
    const TablePtrs& tp = mp.t[blockIdx.y];
    const uint32_t rec_off = mp.rec_off[blockIdx.y];
    const bool lin = tp.lin_in != nullptr;
    const Fr r = load_fr(r_ptr, 0);
    const size_t h = n >> 1, q = n >> 2;
    const uint32_t lane = threadIdx.x & 63, g = threadIdx.x & 3;
    const size_t j = ((size_t)blockIdx.x * MLE_BLOCK + threadIdx.x) >> 2;
    const bool act = j < q;
    Fr mine = Fr::zero(), lmine = Fr::zero();
    if (act) {
        const size_t o = j + ((g & 1) ? q : 0);
        mine = fold_pair(load_fr(tp.in[g >> 1], o), load_fr(tp.in[g >> 1], o + h), r);
        store_fr(tp.out[g >> 1], o, mine);
        if (lin && g < 2) {
            lmine = fold_pair(load_fr(tp.lin_in, o), load_fr(tp.lin_in, o + h), r);
            store_fr(tp.lin_out, o, lmine);
        }
    }
    const int base = (int)(lane & ~3u);
    const Fr lo0 = shfl_fr(mine, base), hi0 = shfl_fr(mine, base + 1), lo1 = shfl_fr(mine, base + 2), hi1 = shfl_fr(mine, base + 3);
    Fr e = Fr::zero();
    if (act && g < 3) {
        const Fr x = g == 0 ? lo0 : g == 1 ? hi0 : hi0 + hi0 - lo0;
        const Fr y = g == 0 ? lo1 : g == 1 ? hi1 : hi1 + hi1 - lo1;
        e = x * y;
    }
    if (lin) {
        const Fr llo = shfl_fr(lmine, base), lhi = shfl_fr(lmine, base + 1);
        if (act && g < 3) e = e + (g == 0 ? llo : g == 1 ? lhi : lhi + lhi - llo);
    }
#pragma unroll
    for (int d = 32; d >= 4; d >>= 1) e = e + shfl_down_fr(e, d);   // lanes 0..2: the wave's sums for t = lane
    if (lane < 3) red[lane][threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x < 3) {
        Fr s = red[threadIdx.x][0];
        for (int w = 1; w < MLE_BLOCK / 64; ++w) s = s + red[threadIdx.x][w];
        store_fr(partials, (size_t)blockIdx.x * rec + rec_off + threadIdx.x, s);
    }
}

// sums the n_partials workgroup records of a round (wave w takes the values v = w, w + 4, ...) and closes it
static __global__ __launch_bounds__(MLE_BLOCK) void composed_close_kernel(const uint64_t* __restrict__ partials, uint32_t n_partials,
                                                                   CloseArgs ca) {
    if (ca.outer.dev && blockIdx.x == gridDim.x - 1) { outer_absorb_rounds(ca.outer, ca.round, 1); return; }   // the hasher workgroup
    __shared__ CloseShared sh;
    __shared__ Sha256State trs;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ZK_STAMP_AT(0, ca.round, 6);
    // the transcript state and the interpolation entries travel to LDS while the records are summed
    if (ca.first != 1 && threadIdx.x < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&trs)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&ca.st->transcript)[threadIdx.x];
    close_preload(sh, ca.meta, ca.st);
    // wave w sums the values v = w, w + 4, ... of the records; the loads of all its values are issued before the first reduction
    constexpr int PER_WAVE = (CMP_MAX_REC + MLE_BLOCK / 64 - 1) / (MLE_BLOCK / 64);
    Fr s[PER_WAVE];
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
        const uint32_t v = wave + q * (MLE_BLOCK / 64);
        s[q] = Fr::zero();
        if (v < ca.meta.rec)
            for (uint32_t b = lane; b < n_partials; b += 256) {   // four loads in flight per lane (up to 2048 records: 8 trips instead of 32)
                Fr x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = b + 64 * u < n_partials ? load_fr(partials, (size_t)(b + 64 * u) * ca.meta.rec + v) : Fr::zero();
                s[q] = s[q] + ((x[0] + x[1]) + (x[2] + x[3]));
            }
    }
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
        const uint32_t v = wave + q * (MLE_BLOCK / 64);
        if (v < ca.meta.rec) {
            const Fr t = wave_reduce_fr(s[q]);
            if (lane == 0) sh.evals[v] = t;
        }
    }
    __syncthreads();
    close_round(sh, ca, &trs, ca.round, ca.first);
    if (threadIdx.x < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&ca.st->transcript)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&trs)[threadIdx.x];
}

// the records of a round summed into one (what a rank contributes to the exchange of the sharded protocol)
static __global__ __launch_bounds__(MLE_BLOCK) void composed_reduce_kernel(const uint64_t* __restrict__ partials, uint32_t n_partials,
                                                                    uint32_t rec, uint64_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t v = wave; v < rec; v += MLE_BLOCK / 64) {
        Fr s = Fr::zero();
        for (uint32_t b = lane; b < n_partials; b += 256) {   // four loads in flight per lane, as in composed_close_kernel
            Fr x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = b + 64 * u < n_partials ? load_fr(partials, (size_t)(b + 64 * u) * rec + v) : Fr::zero();
            s = s + ((x[0] + x[1]) + (x[2] + x[3]));
        }
        s = wave_reduce_fr(s);
        if (lane == 0) store_fr(out, v, s);
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

// ---- the last rounds in one launch ------------------------------------------------------------------------
// Once all tables of the claim fit the LDS (CMP_TAIL_ENTRIES field elements, 128 KiB) the remaining rounds run inside
// one workgroup: tables are loaded once (folded at the previous challenge on the way in), every round evaluates the
// terms' sums from LDS, closes the round (close_round) and folds in place.  A 2^20 claim spends half of its rounds here.
constexpr int CMP_TAIL_BLOCK = 512;
constexpr uint32_t CMP_TAIL_ENTRIES = 4096;
struct TailTables {
    const uint64_t* in[CMP_MAX_TERMS * (CMP_MAX_K + 1)];   // the terms' product tables, then the additive tables
};
inline uint32_t composed_tail_len(uint32_t total_tables) {   // entries per table the tail can hold
    uint32_t m = 1;
    while (2 * m * total_tables <= CMP_TAIL_ENTRIES) m *= 2;
    return m;
}
// one term: per-wave sums of the K+1 evaluations over the pairs (j, j + cn/2) of its K tables (table k at tab + k * m)
template <int K>
__device__ __forceinline__ void tail_term_sums(const uint32_t* tab, const uint32_t* lin /* the term's additive table or nullptr */,
                                               uint32_t m, uint32_t cn, Fr* wave_part /* [rec] of this wave */, uint32_t rec_off) {
    Fr sums[K + 1];
#pragma unroll
    for (int t = 0; t <= K; ++t) sums[t] = Fr::zero();
    const uint32_t half = cn >> 1;
    if ((threadIdx.x & ~63u) < half) {   // waves without a pair keep their zeros
        for (uint32_t j = threadIdx.x; j < half; j += CMP_TAIL_BLOCK) {
            Fr lo[K], hi[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { lo[k] = lds_load_fr(tab, k * m + j); hi[k] = lds_load_fr(tab, k * m + j + half); }
            accumulate_round_evals<K>(lo, hi, sums);
            if (lin) accumulate_linear_evals<K>(lds_load_fr(lin, j), lds_load_fr(lin, j + half), sums);
        }
#pragma unroll
        for (int t = 0; t <= K; ++t) sums[t] = wave_reduce_fr(sums[t]);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int t = 0; t <= K; ++t) wave_part[rec_off + t] = sums[t];
    }
}

// The tail's work beside the hash: the upper half of every table becomes d^ = to_mont(hi - lo), so that the fold behind the
// round needs the challenge only as the hash yields it -- lo + mont(c, d^) = lo + r (hi - lo) for the canonical c -- and nobody
// waits for the challenge's conversion to Montgomery form (0.95 us per round).
struct TailShadow {
    uint32_t* tab;
    uint32_t total, m, half;   // half = 0: nothing to prepare (last round)
    __device__ __forceinline__ void operator()(uint32_t lane, uint32_t n_lanes) const {
        for (uint32_t idx = lane; idx < total * half; idx += n_lanes) {
            const uint32_t q = idx / half, j = idx % half;
            const Fr lo = lds_load_fr(tab, q * m + j), hi = lds_load_fr(tab, q * m + j + half);
            lds_store_fr(tab, q * m + j + half, fr_to_mont_outlined(hi - lo));
        }
    }
};
// the fold itself, by the waves that do not convert the challenge (1..): tables of `half` entries afterwards
__device__ __forceinline__ void tail_fold(uint32_t* tab, uint32_t total, uint32_t m, uint32_t half, const Fr& c_canon) {
    if (threadIdx.x < 64) return;
    for (uint32_t idx = threadIdx.x - 64; idx < total * half; idx += CMP_TAIL_BLOCK - 64) {
        const uint32_t q = idx / half, j = idx % half;
        const Fr lo = lds_load_fr(tab, q * m + j), dhat = lds_load_fr(tab, q * m + j + half);
        lds_store_fr(tab, q * m + j, lo + fr_mul_outlined(c_canon, dhat));
    }
}
static __global__ __launch_bounds__(CMP_TAIL_BLOCK) void composed_tail_kernel(TailTables tt, uint32_t total, uint32_t m, uint32_t load_fold,
                                                                       const uint64_t* __restrict__ r_ptr, CloseArgs ca,
                                                                       uint32_t n_rounds) {
    if (ca.outer.dev && blockIdx.x == gridDim.x - 1) { outer_absorb_rounds(ca.outer, ca.round, n_rounds); return; }   // the hasher workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(zk_dyn_lds);   // total tables x m elements
    __shared__ CloseShared sh;
    __shared__ Fr wave_part[CMP_TAIL_BLOCK / 64][CMP_MAX_REC];
    __shared__ Sha256State trs;
    {
        Fr r = Fr::zero();
        if (load_fold) r = load_fr(r_ptr, 0);
        for (uint32_t idx = threadIdx.x; idx < total * m; idx += CMP_TAIL_BLOCK) {
            const uint32_t q = idx / m, j = idx % m;
            Fr v = load_fr(tt.in[q], j);
            if (load_fold) v = fold_pair(v, load_fr(tt.in[q], (size_t)j + m), r);
            lds_store_fr(tab, idx, v);
        }
        if (threadIdx.x < sizeof(Sha256State) / 4)
            reinterpret_cast<uint32_t*>(&trs)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&ca.st->transcript)[threadIdx.x];
        close_preload(sh, ca.meta, ca.st);
    }
    __syncthreads();
    uint32_t cn = m, first = ca.first;
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t round = ca.round; round < ca.round + n_rounds; ++round) {
        ZK_STAMP_AT(0, round, 6);
        // Late rounds (all pairs fit one wave): wave w computes evaluation w of the record -- one point t of one term -- for
        // every pair and reduces it, so the critical path is K - 1 products and ONE wave reduction instead of all (K + 1) points
        // of all terms one after another on the same lanes.  Field arithmetic is exact: same sums.
        // Terms of three and more tables take this form from 256 pairs on (a few pairs per lane: K - 1 products each, against
        // (K + 1)(K - 1) per pair with one lane per pair: 24 at K = 5, ~16 us of a round).
        const bool wide_terms = ca.meta.k[0] >= 3 || (ca.meta.n_terms > 1 && ca.meta.k[1] >= 3);
        if ((cn >> 1) <= (wide_terms ? 256u : 64u) && ca.meta.rec <= CMP_TAIL_BLOCK / 64) {
            const uint32_t half = cn >> 1, lane = threadIdx.x & 63;
            if (wave < ca.meta.rec) {
                uint32_t p = 0, q0w = 0;
                while (p + 1 < ca.meta.n_terms && wave >= ca.meta.rec_off[p + 1]) { q0w += ca.meta.k[p]; ++p; }
                const uint32_t t = wave - ca.meta.rec_off[p], K = ca.meta.k[p];
                Fr s = Fr::zero();
                for (uint32_t j = lane; j < half; j += 64) {
                    Fr pr = Fr::zero();
                    for (uint32_t k = 0; k < K; ++k) {
                        const Fr lo = lds_load_fr(tab, (q0w + k) * m + j), hi = lds_load_fr(tab, (q0w + k) * m + j + half);
                        Fr v = t == 0 ? lo : hi;
                        if (t >= 2) {
                            const Fr d = hi - lo;
                            for (uint32_t i = 1; i < t; ++i) v = v + d;
                        }
                        pr = k == 0 ? v : fr_mul_outlined(pr, v);
                    }
                    if (ca.meta.lin_tab[p] != ~0u) {
                        const uint32_t lq = ca.meta.lin_tab[p];
                        const Fr lo = lds_load_fr(tab, lq * m + j), hi = lds_load_fr(tab, lq * m + j + half);
                        Fr v = t == 0 ? lo : hi;
                        if (t >= 2) {
                            const Fr d = hi - lo;
                            for (uint32_t i = 1; i < t; ++i) v = v + d;
                        }
                        pr = pr + v;
                    }
                    s = s + pr;
                }
                if (half >= 64) {
                    s = wave_reduce_fr(s);
                } else {
                    // lanes >= half hold zero: log2(half) shuffle steps instead of six
                    for (uint32_t dd = half > 1 ? (1u << (31 - __builtin_clz(half - 1))) : 0; dd >= 1; dd >>= 1) s = s + shfl_down_fr(s, (int)dd);
                }
                if (lane == 0) sh.evals[wave] = s;
            }
            __syncthreads();
            close_round(sh, ca, &trs, round, first, TailShadow{tab, total, m, cn == 2 ? 0u : half});
            first = 0;
            if (cn == 2) break;
            tail_fold(tab, total, m, half, sh.challenge_canon);
            __syncthreads();
            ZK_STAMP_AT(0, round, 7);
            cn = half;
            continue;
        }
        uint32_t q0 = 0;
        for (uint32_t p = 0; p < ca.meta.n_terms; ++p) {
            const uint32_t* base = tab + 8 * (size_t)q0 * m;
            const uint32_t* lin = ca.meta.lin_tab[p] != ~0u ? tab + 8 * (size_t)ca.meta.lin_tab[p] * m : nullptr;
            switch (ca.meta.k[p]) {
                case 1: tail_term_sums<1>(base, lin, m, cn, wave_part[wave], ca.meta.rec_off[p]); break;
                case 2: tail_term_sums<2>(base, lin, m, cn, wave_part[wave], ca.meta.rec_off[p]); break;
                case 3: tail_term_sums<3>(base, lin, m, cn, wave_part[wave], ca.meta.rec_off[p]); break;
                case 4: tail_term_sums<4>(base, lin, m, cn, wave_part[wave], ca.meta.rec_off[p]); break;
                default: tail_term_sums<5>(base, lin, m, cn, wave_part[wave], ca.meta.rec_off[p]); break;
            }
            q0 += ca.meta.k[p];
        }
        __syncthreads();
        if (threadIdx.x < ca.meta.rec) {
            Fr s = wave_part[0][threadIdx.x];
            for (uint32_t w = 1; w < CMP_TAIL_BLOCK / 64; ++w) s = s + wave_part[w][threadIdx.x];
            sh.evals[threadIdx.x] = s;
        }
        __syncthreads();
        const uint32_t half = cn >> 1;
        close_round(sh, ca, &trs, round, first, TailShadow{tab, total, m, cn == 2 ? 0u : half});
        first = 0;
        if (cn == 2) break;   // the fold after the last round has no consumer
        tail_fold(tab, total, m, half, sh.challenge_canon);
        __syncthreads();
        ZK_STAMP_AT(0, round, 7);
        cn = half;
    }
    // hand the transcript back (a later call may continue this sumcheck's rounds: composed.hip `cont`)
    __syncthreads();
    if (threadIdx.x < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&ca.st->transcript)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&trs)[threadIdx.x];
}

}  // namespace zk

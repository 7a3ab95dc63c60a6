// composed_pipe.hpp -- the last rounds of a composed / multi-composed sumcheck whose terms are products of TWO tables (every GKR
// layer claim, the reference's ComposedSumcheck bench shape), ONE ROUND AHEAD of the transcript.
//
// The reference's round loop (composed_sumcheck.rs:41-57, multi_composed_sumcheck.rs:76-107) is strictly serial through the hash:
// sums -> round polynomial -> SHA-256 -> challenge -> fold -> sums ...  Its sums, however, are POLYNOMIALS in the challenge they wait
// for.  With T the current tables (cn entries, pairs (x, x + cn/2)) and T' = fold(T, c) the tables of the next round,
//     T'[j] = X0 + c (X2 - X0),   T'[j + q] = X1 + c (X3 - X1),        X_b = T[j + b q],  q = cn / 4,  j < q,
// the next round's evaluation of a term A * B (+ L) at t = 0, 1, 2 is
//     e_t(c) = sum_j (P0 + c P1)(Q0 + c Q1) + (L0 + c L1) = F0_t + c F1_t + c^2 F2_t,
//     t = 0: P0 = X0, P1 = X2 - X0;   t = 1: P0 = X1, P1 = X3 - X1;   t = 2: P0 = 2 X1 - X0, P1 = 2 (X3 - X1) - (X2 - X0)
// (Q from B, L from the additive table likewise).  The nine sums per term are computed by the waves that have nothing to do WHILE THE
// HASH OF THE CURRENT ROUND RUNS; when the challenge arrives the next round's sums are two dependent products away (Horner), instead of
// a fold, K + 1 products, a reduction tree and a conversion away.  The three forms are kept in the representation that makes those two
// products end in the CANONICAL integer the transcript absorbs, with the challenge taken as the hash yields it (canonical):
//     F2 as F2 R^2,  F1 as F1 R (Montgomery),  F0 canonical:   mont(F2 R^2, c) = F2 c R;  mont((F1 + F2 c) R, c) = (F1 + F2 c) c;  + F0.
// Interpolation of a degree-2 term (c0 = e0, c2 = (e0 - 2 e1 + e2) / 2, c1 = e1 - e0 - c2) is additions and a halving: valid on canonical
// residues, so no conversion is left on the critical path; the Montgomery forms the caller reads back are produced beside the next hash.
// Field arithmetic is exact: sums, round-polynomial bytes and challenges are those of the round-by-round loop, bit for bit.
// Critical path per round: two products + interpolation + message (~3 us) + the hash, against ~7 us + the hash.
#pragma once
#include "composed_kernels.hpp"
#include "composed_stage.hpp"

namespace zk {

constexpr int PIPE_BLOCK = 768;                       // wave 0: transcript; waves 1..11: message schedules (one block each), output copies, the next round's forms (one group each)
constexpr int PIPE_SHADOW_WAVES = PIPE_BLOCK / 64 - 1;
constexpr int PIPE_OUT_WAVE = 5;                      // (four blocks per round in the GKR shape: waves 1..4 schedule, 6..11 take the six groups)
constexpr int PIPE_GROUPS = CMP_MAX_TERMS * 3;        // (term, t) pairs
constexpr uint32_t PIPE_MAX_Q = 128;                  // forms are computed ahead for tables of <= 4 * PIPE_MAX_Q entries (beyond, they take longer than the hash)

struct PipeShared {
    Fr forms[PIPE_GROUPS][3];        // F0 (canonical), F1 (Montgomery), F2 (x R^2) of the NEXT round's e_t, per (term, t)
    Fr raw[PIPE_GROUPS][5];          // per group: sums of P0 Q0, (P0 + P1)(Q0 + Q1), P1 Q1, L0, L1 (Montgomery), accumulated by the group's wave
    Fr out_canon[CMP_MAX_MONO];      // the round's items as absorbed (canonical), for the Montgomery copies written beside the hash
    uint32_t out_n, out_round;
    Fr prev_challenge;               // canonical challenge of the previous round (its Montgomery form is stored beside the hash)
    uint32_t prev_round, prev_valid;
};

// true when the pipelined kernels take the claim: every term a product of exactly two tables (an additive table allowed)
__host__ __device__ inline bool pipe_eligible(const ComposedMeta& meta) {
    for (uint32_t p = 0; p < meta.n_terms; ++p) if (meta.k[p] != 2) return false;
    return meta.n_terms >= 1;
}

// sum over every aligned segment of `seg` lanes (a power of two <= 64), valid in the LAST lane of the segment: DPP moves on the vector
// ALU (row_shr within rows of 16 lanes, row_bcast across them; lanes without a source contribute zero) -- no LDS crossbar traffic
__device__ __forceinline__ Fr seg_sum_fr(Fr v, uint32_t seg) {
    if (seg >= 2) v = v + dpp_fr<0x111, 0xf>(v);
    if (seg >= 4) v = v + dpp_fr<0x112, 0xf>(v);
    if (seg >= 8) v = v + dpp_fr<0x114, 0xf>(v);
    if (seg >= 16) v = v + dpp_fr<0x118, 0xf>(v);
    if (seg >= 32) v = v + dpp_fr<0x142, 0xa>(v);
    if (seg >= 64) v = v + dpp_fr<0x143, 0xc>(v);
    return v;
}
// one unit of a group.  Products (kind 0 / 1 / 2 = P0 Q0 / (P0 + P1)(Q0 + Q1) / P1 Q1 at index j): one operand per table
__device__ __forceinline__ Fr pipe_operand(const uint32_t* tbl, uint32_t q, uint32_t t, uint32_t kind, uint32_t j) {
    const Fr x0 = lds_load_fr(tbl, j), x1 = lds_load_fr(tbl, j + q), x2 = lds_load_fr(tbl, j + 2 * q), x3 = lds_load_fr(tbl, j + 3 * q);
    const Fr d0 = x2 - x0, d1 = x3 - x1;
    const Fr p0 = t == 0 ? x0 : t == 1 ? x1 : (x1 + x1) - x0;
    const Fr p1 = t == 0 ? d0 : t == 1 ? d1 : (d1 + d1) - d0;
    return kind == 0 ? p0 : kind == 2 ? p1 : p0 + p1;
}
// the additive table's (L0, L1) at index j
struct FrPair { Fr a, b; };
__device__ __forceinline__ FrPair pipe_lin(const uint32_t* tl, uint32_t q, uint32_t t, uint32_t j) {
    const Fr x0 = lds_load_fr(tl, j), x1 = lds_load_fr(tl, j + q), x2 = lds_load_fr(tl, j + 2 * q), x3 = lds_load_fr(tl, j + 3 * q);
    const Fr d0 = x2 - x0, d1 = x3 - x1;
    FrPair r;
    r.a = t == 0 ? x0 : t == 1 ? x1 : (x1 + x1) - x0;
    r.b = t == 0 ? d0 : t == 1 ? d1 : (d1 + d1) - d0;
    return r;
}
// The forms of the NEXT round for group g = (term p, point t) from the LDS tables `tab` (table q at tab + 8 q m, cn entries in use),
// computed by ONE wave (all 64 lanes must call).  a_tab / b_tab / l_tab: table numbers (l_tab = ~0u: none).  Every raw sum is written
// exactly once, by the last lane of the segment (or wave) that summed it.
__device__ __forceinline__ void pipe_group_forms(const uint32_t* tab, uint32_t m, uint32_t cn, uint32_t a_tab, uint32_t b_tab, uint32_t l_tab,
                                                 uint32_t t, Fr (&raw)[5], Fr (&forms)[3]) {
    const uint32_t lane = threadIdx.x & 63, q = cn >> 2, log_q = 31 - __builtin_clz(q);
    const bool has_lin = l_tab != ~0u;
    const uint32_t kinds = has_lin ? 4u : 3u;
    const uint32_t* ta = tab + 8 * (size_t)a_tab * m;
    const uint32_t* tb = tab + 8 * (size_t)b_tab * m;
    const uint32_t* tl = tab + 8 * (size_t)(has_lin ? l_tab : 0u) * m;
    if (!has_lin && lane >= 3 && lane < 5) raw[lane] = Fr::zero();
    if (q >= 64) {
        // a kind at a time: every lane adds up its indices, one wave sum per kind
        for (uint32_t kind = 0; kind < 3; ++kind) {
            Fr a = Fr::zero();
            for (uint32_t j = lane; j < q; j += 64) a = a + fr_mul_outlined(pipe_operand(ta, q, t, kind, j), pipe_operand(tb, q, t, kind, j));
            a = seg_sum_fr(a, 64);
            if (lane == 63) raw[kind] = a;
        }
        if (has_lin) {
            Fr a = Fr::zero(), a2 = Fr::zero();
            for (uint32_t j = lane; j < q; j += 64) { const FrPair l = pipe_lin(tl, q, t, j); a = a + l.a; a2 = a2 + l.b; }
            a = seg_sum_fr(a, 64);
            a2 = seg_sum_fr(a2, 64);
            if (lane == 63) { raw[3] = a; raw[4] = a2; }
        }
    } else {
        // several kinds side by side: segments of q lanes
        const uint32_t units = kinds << log_q;
        for (uint32_t base = 0; base < units; base += 64) {
            const uint32_t u = base + lane;
            const bool live = u < units;
            const uint32_t kind = live ? u >> log_q : 0u, j = u & (q - 1);
            Fr v = Fr::zero(), v2 = Fr::zero();
            if (live && kind < 3) v = fr_mul_outlined(pipe_operand(ta, q, t, kind, j), pipe_operand(tb, q, t, kind, j));
            if (live && kind == 3) { const FrPair l = pipe_lin(tl, q, t, j); v = l.a; v2 = l.b; }
            v = seg_sum_fr(v, q);
            if (has_lin) v2 = seg_sum_fr(v2, q);
            if (live && (lane & (q - 1)) == q - 1) {
                raw[kind] = v;
                if (kind == 3) raw[4] = v2;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // lanes read what OTHER lanes of the wave wrote
    // F0 = S0 + L0 (canonical), F1 = Sk - S0 - S2 + L1 (Montgomery), F2 = S2 (x R^2): one lane each
    // (ONE product for the three lanes: x * 1 = x R^-1, x * R = x, x * R^2 = x R in Montgomery arithmetic)
    if (lane < 3) {
        const Fr s0 = raw[0], sk = raw[1], s2 = raw[2], l0 = raw[3], l1 = raw[4];
        const Fr val = lane == 0 ? s0 + l0 : lane == 1 ? ((sk - s0) - s2) + l1 : s2;
        Fr k;
#pragma unroll
        for (int i = 0; i < Fr::N; ++i) k.l[i] = lane == 0 ? (i == 0 ? 1u : 0u) : lane == 1 ? FrParams::r1(i) : FrParams::r2(i);
        forms[lane] = fr_mul_outlined(val, k);
    }
}

// Wave 0, before the hash: the round's evaluations as canonical integers in lanes tid < 3 n_terms -> the items the transcript absorbs
// (sh.canon / sh.pow_of / sh.n_items), their copies for the output conversion (ps.out_canon) and, multi-composed, out[0] = #monomials.
// e: this lane's canonical evaluation (lane 3 p + t).  All 64 lanes of wave 0 must call.
__device__ __forceinline__ void pipe_items_from_evals(CloseShared& sh, PipeShared& ps, const ComposedMeta& meta, const Fr& e, uint32_t round,
                                                      uint64_t* __restrict__ round_out) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t* out = round_out + 64 * (size_t)round;
    if (!meta.multi) {
        // transcript.commit(&vec_to_bytes(&round_poly))  composed_sumcheck.rs:51: the raw evaluations
        if (lane < 3) { sh.canon[lane] = e; ps.out_canon[lane] = e; }
        if (lane == 0) { sh.n_items = 3; ps.out_n = 3; ps.out_round = round; }
        return;
    }
    // round_poly = sum over terms of interpolation(evals at x = 0, 1, 2) (:79-95); coefficients that are zero are dropped per term
    // (sparse_univariate.rs:55) but a zero produced by the sum is kept (:159-203).  c0 = e0, c2 = (e0 - 2 e1 + e2) / 2, c1 = e1 - e0 - c2.
    // The evaluations travel through LDS (one wave: its LDS operations complete in order); lane k < 3 builds power k.
    if (lane < 3 * meta.n_terms) sh.evals[lane] = e;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    Fr sum = Fr::zero();
    bool any = false;
    if (lane < 3) {
        for (uint32_t p = 0; p < meta.n_terms; ++p) {
            const Fr e0 = sh.evals[3 * p], e1 = sh.evals[3 * p + 1], e2 = sh.evals[3 * p + 2];
            Fr coef = e0;
            if (lane > 0) {
                const Fr c2 = fr_half((e0 - (e1 + e1)) + e2);
                coef = lane == 2 ? c2 : (e1 - e0) - c2;
            }
            if (!coef.is_zero()) { sum = sum + coef; any = true; }
        }
    }
    const bool have = lane < 3 && any;
    const uint64_t mask = __ballot(have);
    if (have) {
        const uint32_t at = __popcll(mask & (((uint64_t)1 << lane) - 1));
        sh.canon[at] = sum;
        sh.pow_of[at] = lane;
        ps.out_canon[at] = sum;
        Fr powm = Fr::zero();                            // Fr::from(lane): <= 2 additions of one
        for (uint32_t i = 0; i < lane; ++i) powm = powm + Fr::one();
        store_fr(out + 8, 2 * at + 1, powm);
    }
    if (lane == 0) { sh.n_items = (uint32_t)__popcll(mask); ps.out_n = (uint32_t)__popcll(mask); ps.out_round = round; out[0] = (uint64_t)__popcll(mask); }
}

// The same from the FORMS of the round (wave 0; every lane must call): interpolation is linear, so it is applied to the forms -- lane
// (group (p, k), level l) of four-lane groups holds G_l of coefficient k of term p -- and Horner at the previous challenge then yields
// the coefficients themselves, canonical: G0 + c (G1 + c G2), two wave-wide products with the partial results handed down the group by
// DPP (row_shl:1; a group of four lanes never straddles a row of 16).  ComposedSumcheck (multi == 0) absorbs the raw evaluations: no
// interpolation, k = t.
__device__ __forceinline__ void pipe_items_from_forms(CloseShared& sh, PipeShared& ps, const ComposedMeta& meta, uint32_t round,
                                                      uint64_t* __restrict__ round_out) {
    const uint32_t lane = threadIdx.x & 63, grp = lane >> 2, l = lane & 3;
    const uint32_t n_grp = 3 * meta.n_terms;
    const bool valid = grp < n_grp && l < 3;
    uint64_t* out = round_out + 64 * (size_t)round;
    const uint32_t p = valid ? grp / 3 : 0u, k = valid ? grp - 3 * p : 0u;
    Fr g = Fr::zero();
    if (valid) {
        if (!meta.multi) g = ps.forms[grp][l];
        else {
            const Fr f0 = ps.forms[3 * p][l];
            if (k == 0) g = f0;
            else {
                const Fr f1 = ps.forms[3 * p + 1][l], f2 = ps.forms[3 * p + 2][l];
                const Fr c2 = fr_half((f0 - (f1 + f1)) + f2);
                g = k == 2 ? c2 : (f1 - f0) - c2;
            }
        }
    }
    const Fr c = sh.challenge_canon;                          // the previous round's challenge, as the hash yielded it
    const Fr t1 = fr_mul_outlined(g, c);                      // l == 2: G2 R^2 -> G2 c R
    const Fr s = g + dpp_fr<0x101, 0xf>(t1);                  // l == 1: (G1 + G2 c) R
    const Fr t2 = fr_mul_outlined(s, c);                      // l == 1: (G1 + G2 c) c, canonical
    const Fr coef = g + dpp_fr<0x101, 0xf>(t2);               // l == 0: G0 + c (G1 + c G2), canonical
    if (!meta.multi) {
        if (valid && l == 0) { sh.canon[grp] = coef; ps.out_canon[grp] = coef; }
        if (lane == 0) { sh.n_items = 3; ps.out_n = 3; ps.out_round = round; }
        return;
    }
    // coefficients that are zero are dropped per term (sparse_univariate.rs:55), a zero produced by the sum is kept (:159-203)
    if (valid && l == 0) sh.evals[grp] = coef;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    Fr sum = Fr::zero();
    bool any = false;
    if (lane < 3) {
        for (uint32_t pp = 0; pp < meta.n_terms; ++pp) {
            const Fr v = sh.evals[3 * pp + lane];
            if (!v.is_zero()) { sum = sum + v; any = true; }
        }
    }
    const bool have = lane < 3 && any;
    const uint64_t mask = __ballot(have);
    if (have) {
        const uint32_t at = __popcll(mask & (((uint64_t)1 << lane) - 1));
        sh.canon[at] = sum;
        sh.pow_of[at] = lane;
        ps.out_canon[at] = sum;
        Fr powm = Fr::zero();                            // Fr::from(lane): <= 2 additions of one
        for (uint32_t i = 0; i < lane; ++i) powm = powm + Fr::one();
        store_fr(out + 8, 2 * at + 1, powm);
    }
    if (lane == 0) { sh.n_items = (uint32_t)__popcll(mask); ps.out_n = (uint32_t)__popcll(mask); ps.out_round = round; out[0] = (uint64_t)__popcll(mask); }
}

// the round's padded message (every thread; barrier before and after): what the hasher still holds || [claimed sum] || items || padding
__device__ __forceinline__ void pipe_message(CloseShared& sh, const ComposedMeta& meta, const Sha256State* tr_state, uint32_t first) {
    const uint32_t tid = threadIdx.x;
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
            const uint32_t qq = w - pending - sum_words, i = qq / item_words, o = qq % item_words;
            v = o < 8 ? sh.canon[i].l[7 - o] : (o == 15 ? sh.pow_of[i] : 0u);   // coeff || pow, big-endian (sparse_univariate.rs:27-34)
        } else if (w == body) v = 0x80000000u;
        else if (w == 16 * n_blocks - 2) v = (uint32_t)(bits >> 32);
        else if (w == 16 * n_blocks - 1) v = (uint32_t)bits;
        sh.msg[w] = v;
    }
    if (tid == 0) sh.n_blocks = n_blocks;
    if (tid < CMP_MAX_BLOCKS) sh.kw_ready[tid] = 0;
}

// Wave 0: the state rounds of every block, the challenge (canonical) into sh.challenge_canon, the transcript state into tr_state.
__device__ __forceinline__ void pipe_hash_wave(CloseShared& sh, Sha256State* tr_state, uint32_t first) {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    if (first != 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = tr_state->h[i];
    }
    const uint32_t n_blocks = sh.n_blocks;
    sha256_message_split(h, sh.msg, sh.kw, sh.kw_ready, n_blocks);
    Fr c;                                                   // from_be_bytes_mod_order (fiat_shamir.rs:27-29)
#pragma unroll
    for (int i = 0; i < 8; ++i) c.l[i] = h[7 - i];
    c.reduce_once();
    c.reduce_once();
    if (threadIdx.x == 0) {
        Transcript tr;                                      // finalize_reset + update(digest)
        tr.init();
        tr.commit_words8(h);
        tr.store(tr_state);
        sh.challenge_canon = c;
    }
}
// Waves 1..: the message schedules (one block per wave, round robin)
__device__ __forceinline__ void pipe_schedules(CloseShared& sh) {
    const uint32_t wave = (threadIdx.x >> 6) - 1, n_sched = (blockDim.x >> 6) - 1;
    for (uint32_t b0 = 4 * wave; b0 < sh.n_blocks; b0 += 4 * n_sched) {    // a block per row of 16 lanes
        const uint32_t b = b0 + ((threadIdx.x >> 4) & 3);
        const bool active = b < sh.n_blocks;
        const uint32_t bb = active ? b : b0;
        sha256_schedule_rows_to_lds(sh.msg[16 * bb + (threadIdx.x & 15)], sh.kw + 64 * bb, &sh.kw_ready[bb], 0u, bb == 0 ? 1u : 0u, active);
    }
}
// Beside the hash (lanes of waves 1..): the Montgomery copies the caller reads back -- the round's items and the previous challenge
__device__ __forceinline__ void pipe_outputs(const PipeShared& ps, const ComposedMeta& meta, uint64_t* __restrict__ round_out,
                                             uint64_t* __restrict__ challenges) {
    if (threadIdx.x < 64 * PIPE_OUT_WAVE) return;
    const uint32_t sl = threadIdx.x - 64 * PIPE_OUT_WAVE;   // lane of the wave that has neither a schedule nor a group in the common shapes
    if (sl < ps.out_n) {
        uint64_t* out = round_out + 64 * (size_t)ps.out_round;
        const Fr mont = fr_to_mont_outlined(ps.out_canon[sl]);
        if (meta.multi) store_fr(out + 8, 2 * sl, mont);
        else store_fr(out, sl, mont);
    }
    if (sl == 32 && ps.prev_valid) store_fr(challenges, ps.prev_round, fr_to_mont_outlined(ps.prev_challenge));
}

// sums the forms records of n_records workgroups into ps.forms and fetches the challenge they are to be evaluated at -- the one of the
// round before `round`, the round they belong to (every thread calls)
__device__ __forceinline__ void pipe_reduce_records(CloseShared& sh, PipeShared& ps, const ComposedDev* st, const uint64_t* __restrict__ records,
                                                    uint32_t n_records, uint32_t n_groups, uint32_t round) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_waves = blockDim.x >> 6, vals = 3 * n_groups;
    if (tid < 8) sh.challenge_canon.l[tid] = reinterpret_cast<const uint32_t*>(st->last_canon[(round - 1) & 1])[tid];
    for (uint32_t v = wave; v < vals; v += n_waves) {
        Fr s = Fr::zero();
        for (uint32_t b = lane; b < n_records; b += 256) {       // four loads in flight per lane
            Fr x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = b + 64 * u < n_records ? load_fr(records, (size_t)(b + 64 * u) * vals + v) : Fr::zero();
            s = s + ((x[0] + x[1]) + (x[2] + x[3]));
        }
        s = seg_sum_fr(s, 64);
        if (lane == 63) ps.forms[v / 3][v % 3] = s;
    }
}

// All remaining rounds of a claim whose tables fit the LDS, one round ahead of the transcript (see the head of this file).
// Arguments as composed_tail_kernel.
// records_in (n_records_in > 0): the launch continues the pipelined rounds on larger tables (composed_pipe_round_kernel) -- the m entries
// per table are those of the last closed round, still to be folded at its challenge (ca.st->last_canon), and the records hold the forms
// of the round this launch closes first.
static __global__ __launch_bounds__(PIPE_BLOCK) void composed_tail_pipe_kernel(TailTables tt, uint32_t total, uint32_t m, uint32_t load_fold,
                                                                         const uint64_t* __restrict__ r_ptr, CloseArgs ca, uint32_t n_rounds,
                                                                         const uint64_t* __restrict__ records_in, uint32_t n_records_in, uint32_t max_q) {
    if (ca.outer.dev && blockIdx.x == gridDim.x - 1) { outer_absorb_rounds(ca.outer, ca.round, n_rounds); return; }   // the hasher workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    uint32_t* tab = reinterpret_cast<uint32_t*>(zk_dyn_lds);   // total tables x m elements
    __shared__ CloseShared sh;
    __shared__ PipeShared ps;
    __shared__ Fr wave_part[CMP_TAIL_BLOCK / 64][CMP_MAX_REC];
    __shared__ Sha256State trs;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const ComposedMeta& meta = ca.meta;
    {
        Fr r = Fr::zero();
        if (load_fold) r = load_fr(r_ptr, 0);
        for (uint32_t idx = tid; idx < total * m; idx += PIPE_BLOCK) {
            const uint32_t q = idx / m, j = idx % m;
            Fr v = load_fr(tt.in[q], j);
            if (load_fold) v = fold_pair(v, load_fr(tt.in[q], (size_t)j + m), r);
            lds_store_fr(tab, idx, v);
        }
        if (tid < sizeof(Sha256State) / 4)
            reinterpret_cast<uint32_t*>(&trs)[tid] = reinterpret_cast<const uint32_t*>(&ca.st->transcript)[tid];
        if (tid == 0) { ps.prev_valid = 0; ps.out_n = 0; }
        if (meta.multi && ca.first && tid == 64) sh.sum_canon = fr_from_mont_outlined(close_claimed_sum(ca));   // multi_composed_sumcheck.rs:70
    }
    const uint32_t n_groups = 3 * meta.n_terms;
    if (n_records_in) pipe_reduce_records(sh, ps, ca.st, records_in, n_records_in, n_groups, ca.round);
    __syncthreads();
    uint32_t cn = m, first = ca.first;
    bool have_forms = false, pending_fold = false;
    if (n_records_in) {
        have_forms = true; pending_fold = true; cn = m >> 1;
        if (tid == 64) store_fr(ca.challenges, ca.round - 1, fr_to_mont_outlined(sh.challenge_canon));   // whoever folds by a challenge files its Montgomery form
    }
    for (uint32_t round = ca.round; round < ca.round + n_rounds; ++round) {
        const uint32_t half = cn >> 1;
        ZK_STAMP_AT(0, round, 6);
        Fr e = Fr::zero();
        if (!have_forms) {
            // no forms for this round (the first round of the launch, and the rounds whose forms would take longer than the hash): the
            // round-by-round way -- fold at the previous challenge, then the sums themselves (as composed_tail_kernel computes them)
            if (pending_fold) {
                const Fr cm = fr_to_mont_outlined(sh.challenge_canon);
                for (uint32_t idx = tid; idx < total * cn; idx += PIPE_BLOCK) {
                    const uint32_t q = idx / cn, j = idx % cn;
                    const Fr lo = lds_load_fr(tab, q * m + j), hi = lds_load_fr(tab, q * m + j + cn);
                    lds_store_fr(tab, q * m + j, lo + fr_mul_outlined(cm, hi - lo));
                }
                __syncthreads();
            }
            for (uint32_t p = 0; p < meta.n_terms; ++p) {
                const uint32_t* base = tab + 8 * (size_t)(2 * p) * m;
                const uint32_t* lin = meta.lin_tab[p] != ~0u ? tab + 8 * (size_t)meta.lin_tab[p] * m : nullptr;
                if (tid < CMP_TAIL_BLOCK) tail_term_sums<2>(base, lin, m, cn, wave_part[wave], meta.rec_off[p]);     // (strides by CMP_TAIL_BLOCK threads)
            }
            __syncthreads();
            if (tid < meta.rec) {
                Fr s = wave_part[0][tid];
                for (uint32_t w = 1; w < CMP_TAIL_BLOCK / 64; ++w) s = s + wave_part[w][tid];
                e = fr_from_mont_outlined(s);
            }
        } else if (wave != 0 && pending_fold) {
            // beside wave 0's evaluation of the forms: the tables of this round (nobody on the critical path waits for them)
            const Fr cm = fr_to_mont_outlined(sh.challenge_canon);
            for (uint32_t idx = tid - 64; idx < total * cn; idx += PIPE_BLOCK - 64) {
                const uint32_t q = idx / cn, j = idx % cn;
                const Fr lo = lds_load_fr(tab, q * m + j), hi = lds_load_fr(tab, q * m + j + cn);
                lds_store_fr(tab, q * m + j, lo + fr_mul_outlined(cm, hi - lo));
            }
        }
        ZK_STAMP_AT(0, round, 0);
        if (wave == 0) {
            if (have_forms) pipe_items_from_forms(sh, ps, meta, round, ca.round_out);     // two products away from the items
            else pipe_items_from_evals(sh, ps, meta, e, round, ca.round_out);
        }
        __syncthreads();
        ZK_STAMP_AT(0, round, 1);
        pipe_message(sh, meta, &trs, first);
        __syncthreads();
        ZK_STAMP_AT(0, round, 2);
        // ---- the hash (wave 0) | schedules, output conversions and the NEXT round's forms (waves 1..)
        const bool last = cn == 2 || round + 1 == ca.round + n_rounds;
        const bool make_forms = !last && cn >= 4 && (cn >> 2) <= max_q;
        if (wave == 0) {
            ZK_STAMP_AT(0, round, 3);
            pipe_hash_wave(sh, &trs, first);
            ZK_STAMP_AT(0, round, 4);
        } else {
            pipe_schedules(sh);
            if (wave == PIPE_OUT_WAVE) outer_publish(ca.outer, sh, round);
            pipe_outputs(ps, meta, ca.round_out, ca.challenges);
            if (make_forms) {
                // the tables of this round are folded (above) -- their four quarter blocks give the next round's forms in THIS round's challenge
                // the waves without a message schedule to compute take the groups first
                for (uint32_t g = PIPE_SHADOW_WAVES - wave; g < n_groups; g += PIPE_SHADOW_WAVES) {
                    const uint32_t p = g / 3, t = g - 3 * p;
                    pipe_group_forms(tab, m, cn, 2 * p, 2 * p + 1, meta.lin_tab[p], t, ps.raw[g], ps.forms[g]);
                }
            }
            ZK_STAMP_AT(64, round, 7);
        }
        __syncthreads();
        ZK_STAMP_AT(0, round, 5);
        first = 0;
        if (tid == 0) { ps.prev_challenge = sh.challenge_canon; ps.prev_round = round; ps.prev_valid = 1; }
        if (last) break;                    // the fold after the last round has no consumer
        have_forms = make_forms;
        pending_fold = true;
        cn = half;
    }
    // the last challenge's Montgomery form, and the transcript back to the context (a later call may continue this sumcheck: `cont`)
    __syncthreads();
    if (tid == 0 && ps.prev_valid) store_fr(ca.challenges, ps.prev_round, fr_to_mont_outlined(ps.prev_challenge));
    if (tid < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&ca.st->transcript)[tid] = reinterpret_cast<const uint32_t*>(&trs)[tid];
}


// ---- the rounds on tables too large for one workgroup's LDS, one round ahead as well ---------------------------------------------------
// ONE launch per round (the round-by-round form takes two: sums, then the closing kernel that waits for them).  Workgroup 0 closes round r
// from the forms the previous launch left (Horner at challenge r - 1, interpolation, transcript); the other workgroups fold the tables at
// challenge r - 1 and compute, from the folded tables' four quarter blocks, the forms of round r + 1 in challenge r -- which workgroup 0
// is only now deriving.  Nothing in a launch waits for anything else in it; the next launch finds both results.
struct PipeRoundArgs {
    CloseArgs ca;                  // ca.round: the round workgroup 0 closes (do_close)
    MultiTablePtrs tabs;           // per term: in[0], in[1], lin_in = the source tables; out[..] / lin_out = the folded tables (fold)
    size_t cn;                     // entries per table AFTER the fold (= of the source when fold == 0)
    uint32_t fold;                 // 1: tables = fold(source, challenge of round ca.round - 1); its Montgomery form goes to challenges[fold_round]
    uint32_t fold_round;
    uint32_t do_close;             // workgroup 0 closes round ca.round: 1 = from the forms in records_in; 2 = from the sums themselves (records of
                                   // meta.rec Montgomery values per workgroup, as the round kernels of composed_kernels.hpp leave them: the first
                                   // round of a pipelined stretch); 0 = no closing workgroup
    const uint64_t* records_in;
    uint32_t n_records_in;
    uint64_t* records_out;         // one record of 9 n_terms forms per cross workgroup
};
// indices j per tile: 4 x PIPE_TILE entries of every table in LDS.  A tile is what ONE workgroup takes through both phases in a row:
// 13.5 us at 64 (two entries to fold per thread, two passes of jobs per wave), 7.7 us at 32 (in-kernel stamps, tools/diag_composed.py) --
// beside a closing workgroup that needs 12 us.  End to end the two measure the same (tools/ab_tile.sh, same box, three runs each:
// ComposedSumcheck 2^22 0.565-0.591 / 0.585-0.604 ms, GKR depth 20 10.08-10.12 / 9.97-10.38 ms at 64 / 32).
#ifndef ZK_PIPE_TILE
#define ZK_PIPE_TILE 64
#endif
constexpr uint32_t PIPE_TILE = ZK_PIPE_TILE;
static_assert(PIPE_TILE == 16 || PIPE_TILE == 32 || PIPE_TILE == 64, "a job is an aligned segment of a wave");

// The closing half of a round, from sh.evals-free state: ps.forms (from_forms) or the canonical sums e (per thread < meta.rec) are in place.
// Every thread of the workgroup calls; the challenge ends in sh.challenge_canon, the transcript in trs.
struct PipeNothing { __device__ __forceinline__ void operator()() const {} };
template <class Beside = PipeNothing>
__device__ __forceinline__ void pipe_close_round(CloseShared& sh, PipeShared& ps, Sha256State& trs, const CloseArgs& ca, uint32_t round, uint32_t first,
                                                 bool from_forms, const Fr& e, Beside beside = Beside()) {
    const uint32_t wave = threadIdx.x >> 6;
    ZK_STAMP_AT(0, round, 6);
    ZK_STAMP_AT(0, round, 0);
    if (wave == 0) {
        if (from_forms) pipe_items_from_forms(sh, ps, ca.meta, round, ca.round_out);
        else pipe_items_from_evals(sh, ps, ca.meta, e, round, ca.round_out);
    }
    __syncthreads();
    ZK_STAMP_AT(0, round, 1);
    pipe_message(sh, ca.meta, &trs, first);
    __syncthreads();
    ZK_STAMP_AT(0, round, 2);
    if (wave == 0) {
        ZK_STAMP_AT(0, round, 3);
        pipe_hash_wave(sh, &trs, first);
        ZK_STAMP_AT(0, round, 4);
    } else {
        pipe_schedules(sh);
        if (wave == PIPE_OUT_WAVE) outer_publish(ca.outer, sh, round);   // (the wave without a schedule in the common shapes)
        pipe_outputs(ps, ca.meta, ca.round_out, ca.challenges);
        beside();                                    // (waves 1..: whatever else can be prepared while wave 0 hashes)
    }
    __syncthreads();
    ZK_STAMP_AT(0, round, 5);
    if (threadIdx.x < 8) reinterpret_cast<uint32_t*>(ca.st->last_canon[round & 1])[threadIdx.x] = sh.challenge_canon.l[threadIdx.x];
}

// source / destination of every table slot (3 p + {0, 1, 2}: the term's two factors and its additive table; null = no table there)
struct PipeTileTabs {
    const uint64_t* src[PIPE_GROUPS];
    uint64_t* dst[PIPE_GROUPS];
};
__device__ __forceinline__ Fr pipe_challenge_mont(const ComposedDev* st, uint32_t round) {      // the challenge of `round`, Montgomery form
    Fr c;
#pragma unroll
    for (int i = 0; i < Fr::N; ++i) c.l[i] = reinterpret_cast<const uint32_t*>(st->last_canon[round & 1])[i];
    return fr_to_mont_outlined(c);
}
// One tile of PIPE_TILE indices j: the tile's 4 quarter-block entries of every table into LDS -- folded by cm on the way in and written to
// dst (fold) -- then the products of the next round's forms, accumulated into ps.raw.  Every thread of the workgroup calls.
// cm_src != nullptr: cm is still to be fetched (the challenge of round cm_round, canonical in the context) -- it is, AFTER the tile's loads
// are issued: the loads do not depend on it, and its own load and conversion run in their shadow (~1 us per launch).
__device__ __forceinline__ void pipe_cross_tile(const PipeTileTabs& tb, uint32_t n_terms, size_t cn, bool fold, Fr& cm, const ComposedDev*& cm_src,
                                                uint32_t cm_round, size_t tile_i, uint32_t* tile, PipeShared& ps) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t n_slots = 3 * n_terms, n_groups = 3 * n_terms;
    const size_t q = cn >> 2, j0 = tile_i * PIPE_TILE;
    const uint32_t J = (uint32_t)((q - j0) < PIPE_TILE ? (q - j0) : PIPE_TILE);
    __syncthreads();                                                // the previous tile's products are done with the LDS tile
    // phase 1: the tile of every table, folded on the way in (all loads of a lane are issued before its first product)
    const uint32_t units = n_slots * 4 * PIPE_TILE;
    for (uint32_t u0 = tid; u0 < units; u0 += 2 * PIPE_BLOCK) {
        Fr lo[2], hi[2];
        bool live[2];
        uint32_t slot[2], b[2], jj[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t u = u0 + k * PIPE_BLOCK;
            slot[k] = u / (4 * PIPE_TILE); b[k] = (u / PIPE_TILE) & 3; jj[k] = u & (PIPE_TILE - 1);
            const uint64_t* src = u < units ? tb.src[slot[k]] : nullptr;
            live[k] = src != nullptr && jj[k] < J;
            lo[k] = hi[k] = Fr::zero();
            if (live[k]) {
                const size_t x = j0 + jj[k] + (size_t)b[k] * q;
                lo[k] = load_fr(src, x);
                if (fold) hi[k] = load_fr(src, x + cn);
            }
        }
        if (cm_src) { cm = pipe_challenge_mont(cm_src, cm_round); cm_src = nullptr; }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t u = u0 + k * PIPE_BLOCK;
            if (u >= units) continue;
            Fr v = lo[k];
            if (live[k] && fold) {
                v = lo[k] + fr_mul_outlined(cm, hi[k] - lo[k]);
                store_fr(tb.dst[slot[k]], j0 + jj[k] + (size_t)b[k] * q, v);
            }
            lds_store_fr(tile, (slot[k] * 4 + b[k]) * PIPE_TILE + jj[k], v);       // lanes past the table hold zero
        }
    }
    __syncthreads();
    // phase 2: jobs (kind, group (p, t)) of PIPE_TILE lanes each, 64 / PIPE_TILE of them side by side in a wave: one product per index and a
    // segment sum each (kind 3: the additive table).  Kind-major numbering: the jobs a wave holds are of one kind when n_groups is even.
    constexpr uint32_t PACK = 64 / PIPE_TILE;
    const uint32_t sub = lane / PIPE_TILE, jl = lane & (PIPE_TILE - 1), n_jobs = 4 * n_groups;
    for (uint32_t job0 = wave * PACK; job0 < n_jobs; job0 += (PIPE_BLOCK / 64) * PACK) {
        const uint32_t job = job0 + sub;
        if (job >= n_jobs) continue;
        const uint32_t kind = job / n_groups, g = job - kind * n_groups, p = g / 3, t = g - 3 * p;
        if (kind < 3) {
            const uint32_t* ta = tile + 8 * (size_t)(3 * p) * 4 * PIPE_TILE;
            const uint32_t* tbb = tile + 8 * (size_t)(3 * p + 1) * 4 * PIPE_TILE;
            const Fr v = seg_sum_fr(fr_mul_outlined(pipe_operand(ta, PIPE_TILE, t, kind, jl), pipe_operand(tbb, PIPE_TILE, t, kind, jl)), PIPE_TILE);
            if (jl == PIPE_TILE - 1) ps.raw[g][kind] = ps.raw[g][kind] + v;
        } else if (tb.src[3 * p + 2] != nullptr) {
            const uint32_t* tl = tile + 8 * (size_t)(3 * p + 2) * 4 * PIPE_TILE;
            const FrPair l = pipe_lin(tl, PIPE_TILE, t, jl);
            const Fr v = seg_sum_fr(l.a, PIPE_TILE), v2 = seg_sum_fr(l.b, PIPE_TILE);
            if (jl == PIPE_TILE - 1) { ps.raw[g][3] = ps.raw[g][3] + v; ps.raw[g][4] = ps.raw[g][4] + v2; }
        }
    }
}
// the workgroup's record from ps.raw: F0 canonical, F1 Montgomery, F2 x R^2 (one product for all)
__device__ __forceinline__ void pipe_write_record(const PipeShared& ps, uint32_t n_groups, uint64_t* __restrict__ records, size_t slot) {
    const uint32_t tid = threadIdx.x;
    if (tid < 3 * n_groups) {
        const uint32_t g = tid / 3, l = tid - 3 * g;
        const Fr s0 = ps.raw[g][0], sk = ps.raw[g][1], s2 = ps.raw[g][2], l0 = ps.raw[g][3], l1 = ps.raw[g][4];
        const Fr val = l == 0 ? s0 + l0 : l == 1 ? ((sk - s0) - s2) + l1 : s2;
        Fr k;
#pragma unroll
        for (int i = 0; i < Fr::N; ++i) k.l[i] = l == 0 ? (i == 0 ? 1u : 0u) : l == 1 ? FrParams::r1(i) : FrParams::r2(i);
        store_fr(records, slot * 3 * n_groups + tid, fr_mul_outlined(val, k));
    }
}

static __global__ __launch_bounds__(PIPE_BLOCK) void composed_pipe_round_kernel(PipeRoundArgs a) {
    // the hasher workgroup (the last one, when an outer transcript is fed and this launch closes a round)
    const uint32_t n_outer = (a.ca.outer.dev && a.do_close) ? 1u : 0u;
    if (n_outer && blockIdx.x == gridDim.x - 1) { outer_absorb_rounds(a.ca.outer, a.ca.round, 1); return; }
    extern __shared__ __attribute__((aligned(16))) unsigned char zk_dyn_lds[];
    __shared__ CloseShared sh;
    __shared__ PipeShared ps;
    __shared__ Sha256State trs;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const ComposedMeta& meta = a.ca.meta;
    const uint32_t n_groups = 3 * meta.n_terms;
    if (a.do_close && blockIdx.x == 0) {
        // ---- workgroup 0: close round ca.round from the forms
        const uint32_t round = a.ca.round;
        if (a.ca.first != 1 && tid < sizeof(Sha256State) / 4)
            reinterpret_cast<uint32_t*>(&trs)[tid] = reinterpret_cast<const uint32_t*>(&a.ca.st->transcript)[tid];
        if (tid == 0) { ps.prev_valid = 0; ps.out_n = 0; }
        const uint32_t first = a.ca.first;
        Fr e = Fr::zero();
        if (a.do_close == 1) {
            pipe_reduce_records(sh, ps, a.ca.st, a.records_in, a.n_records_in, n_groups, round);
            __syncthreads();
        } else {
            if (meta.multi && first && tid == 64) sh.sum_canon = fr_from_mont_outlined(close_claimed_sum(a.ca));   // multi_composed_sumcheck.rs:70
            for (uint32_t v = wave; v < meta.rec; v += PIPE_BLOCK / 64) {
                Fr s = Fr::zero();
                for (uint32_t b = lane; b < a.n_records_in; b += 256) {
                    Fr x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = b + 64 * u < a.n_records_in ? load_fr(a.records_in, (size_t)(b + 64 * u) * meta.rec + v) : Fr::zero();
                    s = s + ((x[0] + x[1]) + (x[2] + x[3]));
                }
                s = seg_sum_fr(s, 64);
                if (lane == 63) sh.evals[v] = s;
            }
            __syncthreads();
            if (tid < meta.rec) e = fr_from_mont_outlined(sh.evals[tid]);
        }
        pipe_close_round(sh, ps, trs, a.ca, round, first, a.do_close == 1, e);
        if (tid < sizeof(Sha256State) / 4)
            reinterpret_cast<uint32_t*>(&a.ca.st->transcript)[tid] = reinterpret_cast<const uint32_t*>(&trs)[tid];
        return;
    }
    // ---- the other workgroups: fold the tables (or take them as they are) and compute the forms of the round after the one being closed
    uint32_t* tile = reinterpret_cast<uint32_t*>(zk_dyn_lds);          // [slot][4 blocks][PIPE_TILE] field elements; slot 3 p + {0, 1, 2}
    const uint32_t n_close = a.do_close ? 1u : 0u, wg = blockIdx.x - n_close, n_cross = gridDim.x - n_close - n_outer;
    const size_t cn = a.cn, q = cn >> 2;
    Fr cm = Fr::zero();
    const ComposedDev* cm_src = a.fold ? a.ca.st : nullptr;            // fetched inside the first tile, behind its loads
    if (tid < n_groups * 5) (&ps.raw[0][0])[tid] = Fr::zero();
    __shared__ PipeTileTabs tb;                                        // (indexed by slot at run time: LDS, not registers)
    if (tid < PIPE_GROUPS) {
        const uint32_t p = tid / 3, w = tid - 3 * p;
        const bool on = p < meta.n_terms;
        tb.src[tid] = !on ? nullptr : w < 2 ? a.tabs.t[p].in[w] : a.tabs.t[p].lin_in;
        tb.dst[tid] = !on ? nullptr : w < 2 ? a.tabs.t[p].out[w] : a.tabs.t[p].lin_out;
    }
    const size_t n_tiles = (q + PIPE_TILE - 1) / PIPE_TILE;
    if (wg == 0) ZK_STAMP_AT(0, 32 + (a.ca.round & 31), 1);
    for (size_t tile_i = wg; tile_i < n_tiles; tile_i += n_cross)
        pipe_cross_tile(tb, meta.n_terms, cn, a.fold != 0, cm, cm_src, a.ca.round - 1, tile_i, tile, ps);
    if (a.fold && wg == 0 && tid == 0) {                                // whoever folds by a challenge files its Montgomery form
        if (cm_src) cm = pipe_challenge_mont(cm_src, a.ca.round - 1);
        store_fr(a.ca.challenges, a.fold_round, cm);
    }
    __syncthreads();
    if (wg == 0) ZK_STAMP_AT(0, 32 + (a.ca.round & 31), 2);
    pipe_write_record(ps, n_groups, a.records_out, wg);
    if (wg == 0) ZK_STAMP_AT(0, 32 + (a.ca.round & 31), 3);
}

// ---- the serial kernel of a STAGE (two rounds per pass over large tables, composed_stage.hpp) in the same form ---------------------------
// Round 1 closes from the cross sums; round 2's sums are the cross sums BOUND at round 1's challenge -- quadratics in it whose three
// coefficients are plain sums of the cross sums (no products):
//     C'[x][y](r) = C00 + r (C01 + C10 - 2 C00) + r^2 (C00 - C01 - C10 + C11),   C_ab = C[2a + x][2b + y];   L'[x](r) = L[x] + r (L[2 + x] - L[x])
//     e_0 = C'[0][0] + L'[0],   e_1 = C'[1][1] + L'[1],   e_2 = C'[0][0] - 2 (C'[0][1] + C'[1][0]) + 4 C'[1][1] - L'[0] + 2 L'[1].
// They are laid out as forms beside round 1's hash, so round 2 starts two products behind it (composed_stage_close_kernel: a bind three
// products deep, then the sums, then the round-by-round closing: 53 us for the two rounds; here ~35).
static __global__ __launch_bounds__(PIPE_BLOCK) void composed_stage_close_pipe_kernel(const uint64_t* __restrict__ partials, StageArgs sa) {
    if (sa.ca.outer.dev && blockIdx.x == gridDim.x - 1) { outer_absorb_rounds(sa.ca.outer, sa.ca.round, 2); return; }   // the hasher workgroup
    __shared__ CloseShared sh;
    __shared__ PipeShared ps;
    __shared__ Sha256State trs;
    __shared__ Fr vals[CMP_MAX_TERMS][CST_VALS];
    __shared__ Fr rsum[PIPE_BLOCK];
    __shared__ Fr r1m;
    const CloseArgs& ca = sa.ca;
    const uint32_t tid = threadIdx.x;
    const uint32_t P = ca.meta.n_terms, n_vals = P * CST_VALS;
    if (ca.first != 1 && tid < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&trs)[tid] = reinterpret_cast<const uint32_t*>(&ca.st->transcript)[tid];
    if (tid == 0) { ps.prev_valid = 0; ps.out_n = 0; }
    if (ca.meta.multi && ca.first && tid == 64) sh.sum_canon = fr_from_mont_outlined(close_claimed_sum(ca));   // multi_composed_sumcheck.rs:70
    {   // the records: thread (value v, chunk) sums the records chunk, chunk + n_chunks, ... (eight loads in flight); the chunks are added in LDS
        const uint32_t n_chunks = PIPE_BLOCK / n_vals, v = tid % n_vals, chunk = tid / n_vals;
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
    // ---- round 1: p(0), p(1), p(2) of every term from C and L (as composed_stage_close_kernel), canonical for the transcript
    Fr e = Fr::zero();
    if (tid < 3 * P) {
        const uint32_t p = tid / 3, t = tid % 3;
        const Fr* C = vals[p];
        const Fr* L = vals[p] + 16;
        Fr v;
        if (t == 0) v = (C[0] + C[5]) + (L[0] + L[1]);
        else if (t == 1) v = (C[10] + C[15]) + (L[2] + L[3]);
        else {
            const Fr ll = C[0] + C[5], hh = C[10] + C[15], lh = (C[2] + C[7]) + (C[8] + C[13]);
            const Fr hh2 = hh + hh, lh2 = lh + lh;
            v = (ll + (hh2 + hh2)) - lh2;
            const Fr lhi = L[2] + L[3];
            v = v + ((lhi + lhi) - (L[0] + L[1]));
        }
        e = fr_from_mont_outlined(v);
    }
    // beside round 1's hash (the last wave): the forms of round 2, thread (p, t, k)
    auto stage_forms = [&]() {
        const uint32_t u = tid - (PIPE_BLOCK - 64);
        if (tid < PIPE_BLOCK - 64 || u >= 9 * P) return;
        const uint32_t p = u / 9, t = (u / 3) % 3, k = u % 3;
        const Fr* C = vals[p];
        const Fr* L = vals[p] + 16;
        auto Q = [&](uint32_t x, uint32_t y) {            // coefficient k of C'[x][y]
            const Fr c00 = C[4 * x + y], c01 = C[4 * x + 2 + y], c10 = C[4 * (2 + x) + y], c11 = C[4 * (2 + x) + 2 + y];
            if (k == 0) return c00;
            const Fr mixed = c01 + c10;
            return k == 1 ? mixed - (c00 + c00) : (c00 + c11) - mixed;
        };
        auto M = [&](uint32_t x) { return k == 0 ? L[x] : k == 1 ? L[2 + x] - L[x] : Fr::zero(); };
        Fr f;
        if (t == 0) f = Q(0, 0) + M(0);
        else if (t == 1) f = Q(1, 1) + M(1);
        else {
            const Fr q11 = Q(1, 1), q2 = q11 + q11, mixed = Q(0, 1) + Q(1, 0), m1 = M(1);
            f = ((Q(0, 0) + (q2 + q2)) - (mixed + mixed)) + ((m1 + m1) - M(0));
        }
        Fr kk;
#pragma unroll
        for (int i = 0; i < Fr::N; ++i) kk.l[i] = k == 0 ? (i == 0 ? 1u : 0u) : k == 1 ? FrParams::r1(i) : FrParams::r2(i);
        ps.forms[3 * p + t][k] = fr_mul_outlined(f, kk);      // F0 canonical, F1 Montgomery, F2 x R^2
    };
    pipe_close_round(sh, ps, trs, ca, ca.round, ca.first, false, e, stage_forms);
    if (tid == 0) r1m = fr_to_mont_outlined(sh.challenge_canon);
    // ---- round 2 from the forms (sh.challenge_canon = round 1's challenge)
    pipe_close_round(sh, ps, trs, ca, ca.round + 1, 0u, true, Fr::zero());
    if (tid < 4) {
        const Fr r2 = fr_to_mont_outlined(sh.challenge_canon), r1v = r1m;
        const Fr one = Fr::one();
        const Fr f1 = (tid >> 1) ? r1v : one - r1v, f2 = (tid & 1) ? r2 : one - r2;
        store_fr(sa.weights_out, tid, (f1 * f2) * fr_mont_2_32());
        if (tid == 0) { store_fr(ca.challenges, ca.round, r1v); store_fr(ca.challenges, ca.round + 1, r2); }
    }
    if (tid < sizeof(Sha256State) / 4)
        reinterpret_cast<uint32_t*>(&ca.st->transcript)[tid] = reinterpret_cast<const uint32_t*>(&trs)[tid];
}

}  // namespace zk

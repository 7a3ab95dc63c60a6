/* gkr_sparse.c -- GKRProtocol::prove restated on SPARSE CONTAINERS (TEST INFRASTRUCTURE ONLY).
 *
 * gkr.c follows the reference literally: per layer it builds the dense 0/1 wiring tables over (a, b, c)
 * (circuit/src/circuit.rs:59-97: 2^(3l+2) entries) and the dense (b, c) tables of the layer's sumcheck
 * (gkr/src/protocol.rs:64-84: 2^(2l+2) entries) -- which stops at depth 8.  This file runs the SAME loops
 *
 *   gkr/src/protocol.rs:21-117                                   GKRProtocol::prove
 *   gkr/src/utils.rs:12-56                                       generate_layer_one_prove_sumcheck
 *   sumcheck/src/composed/multi_composed_sumcheck.rs:64-121      prove_internal (fold variable 0, evaluate at t = 0..=K,
 *                                                                interpolate per term, drop zero coefficients per term,
 *                                                                absorb coeff||pow, challenge, fold)
 *   polynomial/src/multilinear/evaluation_form.rs:123-159        partial_evaluation(s): out = r*y2 + (1-r)*y1 on (i, i + n/2)
 *   polynomial/src/multilinear/evaluation_form.rs:178-194,235-251  Mul<F>, Add
 *
 * on two containers that hold the same tables without their zeros / without their redundancy:
 *
 *   sp_t  a table of n entries kept as the ascending list of its NON-ZERO entries (one per gate).  Folding variable 0
 *         pairs (i, i + n/2): an output entry is non-zero only if one of its two inputs is, so the fold is a merge of
 *         the list's lower and upper half.  Scaling and adding are entry-wise on the lists.  Exactly the dense result,
 *         entry for entry (absent = 0; r*0 + (1-r)*0 = 0).
 *   ds_t  the outer sum / product of two tables, add_distinct / mul_distinct (evaluation_form.rs:28-52), kept as its two
 *         factors: entry i*nv + j = u[i] (+|*) v[j].  Its fold in variable 0 is the fold of u (while u has more than one
 *         entry; then of v):  r*(u2 + v) + (1-r)*(u1 + v) = (r*u2 + (1-r)*u1) + v  and  r*(u2*v) + (1-r)*(u1*v) =
 *         (r*u2 + (1-r)*u1)*v  hold entry for entry in the field, and field elements have one representation, so the
 *         folded container holds bit for bit what folding the dense table yields.  Entries are formed on demand.
 *
 * The round sums only need the entries of the second container at the (paired) positions where the first is non-zero:
 * O(gates) per round, seconds on one core at depth 20.  This is the reference's algorithm -- (b, c) tables, variable 0 =
 * MSB first, every round over ALL remaining (b, c) variables -- not the two-phase linear-time prover of csrc/gkr.hip
 * (row sums over b with c summed out, then c with b bound): the two share nothing but the field arithmetic.
 *
 * Pinned against gkr.c's dense prover for every depth it reaches (tests/test_oracle_kats.py: the reference's GKR test
 * circuits, Circuit::random(1..8), circuits with shared inputs and mixed gate types), which is itself pinned on the
 * reference's tests (gkr.c header).  Proof bytes remain parity-unpinned in the sense of zkoracle.h.
 */
#include "zkoracle.h"
#include <stdlib.h>
#include <string.h>

/* ---- sp_t: non-zero entries of a table ------------------------------------------------------------------------ */
/* positions and table sizes as 128-bit integers: the wiring table of layer l has 2^(3l + 2) entries -- 2^65 at depth 22 */
typedef unsigned __int128 pos_t;
#define POS_MAX (~(pos_t)0)
typedef struct { pos_t *pos; fr_t *val; size_t len; pos_t n; } sp_t;

static int sp_alloc(sp_t *t, size_t cap, pos_t n) {
    t->pos = (pos_t *)malloc((cap ? cap : 1) * sizeof(pos_t));
    t->val = (fr_t *)malloc((cap ? cap : 1) * sizeof(fr_t));
    t->len = 0;
    t->n = n;
    return (t->pos && t->val) ? 0 : -1;
}
static void sp_free(sp_t *t) { free(t->pos); free(t->val); t->pos = NULL; t->val = NULL; t->len = 0; }

/* partial_evaluation(r, 0) (evaluation_form.rs:123-141 with utils.rs:26-53 at variable_index 0: pairs (i, i + n/2)) */
static int sp_fold(sp_t *t, const fr_t *r) {
    if (t->n < 2) return -1;
    const pos_t half = t->n / 2;
    size_t m = 0;
    while (m < t->len && t->pos[m] < half) ++m;                  /* entries [0, m) lie in the lower half */
    sp_t o;
    if (sp_alloc(&o, t->len, half) != 0) return -1;
    fr_t one, one_minus_r, zero;
    ora_fr_one(&one);
    ora_fr_zero(&zero);
    ora_fr_sub(&one_minus_r, &one, r);
    size_t i = 0, j = m;
    while (i < m || j < t->len) {
        const pos_t pi = i < m ? t->pos[i] : POS_MAX, pj = j < t->len ? t->pos[j] - half : POS_MAX;
        const pos_t p = pi < pj ? pi : pj;
        const fr_t *y1 = &zero, *y2 = &zero;
        if (pi == p) y1 = &t->val[i++];
        if (pj == p) y2 = &t->val[j++];
        fr_t t1, t2;
        ora_fr_mul(&t1, r, y2);                                  /* eval_point * y2 + (1 - eval_point) * y1 */
        ora_fr_mul(&t2, &one_minus_r, y1);
        o.pos[o.len] = p;
        ora_fr_add(&o.val[o.len], &t1, &t2);
        o.len++;
    }
    sp_free(t);
    *t = o;
    return 0;
}

/* partial_evaluations(points, [0; k]) (evaluation_form.rs:143-159) of a copy */
static int sp_folds(sp_t *out, const sp_t *in, const fr_t *pts, size_t k) {
    if (sp_alloc(out, in->len, in->n) != 0) return -1;
    memcpy(out->pos, in->pos, in->len * sizeof(pos_t));
    memcpy(out->val, in->val, in->len * sizeof(fr_t));
    out->len = in->len;
    for (size_t i = 0; i < k; ++i)
        if (sp_fold(out, &pts[i]) != 0) return -1;
    return 0;
}

/* (a * alpha) + (b * beta): Mul<F> (evaluation_form.rs:235-251) then Add (:178-194), entry-wise */
static int sp_scale_add(sp_t *out, const sp_t *a, const fr_t *alpha, const sp_t *b, const fr_t *beta) {
    if (a->n != b->n || sp_alloc(out, a->len + b->len, a->n) != 0) return -1;
    size_t i = 0, j = 0;
    while (i < a->len || j < b->len) {
        const pos_t pi = i < a->len ? a->pos[i] : POS_MAX, pj = j < b->len ? b->pos[j] : POS_MAX;
        const pos_t p = pi < pj ? pi : pj;
        fr_t x, y;
        ora_fr_zero(&x);
        ora_fr_zero(&y);
        if (pi == p) ora_fr_mul(&x, &a->val[i++], alpha);
        if (pj == p) ora_fr_mul(&y, &b->val[j++], beta);
        out->pos[out->len] = p;
        ora_fr_add(&out->val[out->len], &x, &y);
        out->len++;
    }
    return 0;
}

/* ---- ds_t: add_distinct / mul_distinct kept as its factors ------------------------------------------------------ */
typedef struct { fr_t *u, *v, *scratch; size_t nu, nv; int mul; } ds_t;

static int ds_init(ds_t *d, const fr_t *w, size_t w_len, int mul) {
    d->u = (fr_t *)malloc(w_len * sizeof(fr_t));
    d->v = (fr_t *)malloc(w_len * sizeof(fr_t));
    d->scratch = (fr_t *)malloc(w_len * sizeof(fr_t));
    if (!d->u || !d->v || !d->scratch) return -1;
    memcpy(d->u, w, w_len * sizeof(fr_t));
    memcpy(d->v, w, w_len * sizeof(fr_t));
    d->nu = d->nv = w_len;
    d->mul = mul;
    return 0;
}
static void ds_free(ds_t *d) { free(d->u); free(d->v); free(d->scratch); }

/* entry x of the table: evaluation_form.rs:33-35 / :46-48 (self.evaluations[i] (+|*) rhs.evaluations[j] at i*len + j) */
static void ds_at(fr_t *o, const ds_t *d, pos_t x) {
    const fr_t *a = &d->u[(size_t)(x / d->nv)], *b = &d->v[(size_t)(x % d->nv)];
    if (d->mul) ora_fr_mul(o, a, b); else ora_fr_add(o, a, b);
}

/* partial_evaluation(r, 0): the fold of the leading factor (identity in the file header) */
static int ds_fold(ds_t *d, const fr_t *r) {
    fr_t **f = d->nu > 1 ? &d->u : &d->v;
    size_t *n = d->nu > 1 ? &d->nu : &d->nv;
    if (*n < 2) return -1;
    if (ora_mle_partial_evaluation(d->scratch, *f, *n, r, 0) != 0) return -1;
    fr_t *t = *f; *f = d->scratch; d->scratch = t;
    *n /= 2;
    return 0;
}

/* ---- one product term [A, S] of two tables: ComposedMultilinear::new(vec![A, S]) -------------------------------- */
/* multi_composed_sumcheck.rs:81-90: for i in 0..=max_degree (= 2 tables, composed_multilinear.rs:101-103):
 *   p.partial_evaluation(&F::from(i), &0).element_wise_product().iter().sum()
 * = sum over j < n/2 of A_i[j] * S_i[j] with X_i[j] = i*X[j + n/2] + (1 - i)*X[j]; terms with A_i[j] = 0 for every i
 * (both paired entries of A absent) are zero and skipped.  X_{i+1} = X_i + (X[j + n/2] - X[j]). */
static void term_round_evals(fr_t ev[3], const sp_t *a, const ds_t *s) {
    const pos_t half = a->n / 2;
    size_t m = 0;
    while (m < a->len && a->pos[m] < half) ++m;
    fr_t zero;
    ora_fr_zero(&zero);
    for (int t = 0; t < 3; ++t) ora_fr_zero(&ev[t]);
    size_t i = 0, j = m;
    while (i < m || j < a->len) {
        const pos_t pi = i < m ? a->pos[i] : POS_MAX, pj = j < a->len ? a->pos[j] - half : POS_MAX;
        const pos_t p = pi < pj ? pi : pj;
        const fr_t *a1 = &zero, *a2 = &zero;
        if (pi == p) a1 = &a->val[i++];
        if (pj == p) a2 = &a->val[j++];
        fr_t s1, s2, da, dsv, at, st, pr;
        ds_at(&s1, s, p);
        ds_at(&s2, s, p + half);
        ora_fr_sub(&da, a2, a1);
        ora_fr_sub(&dsv, &s2, &s1);
        at = *a1;
        st = s1;
        for (int t = 0; t < 3; ++t) {
            ora_fr_mul(&pr, &at, &st);
            ora_fr_add(&ev[t], &ev[t], &pr);
            ora_fr_add(&at, &at, &da);
            ora_fr_add(&st, &st, &dsv);
        }
    }
}

/* MultiComposedSumcheckProver::prove_partial (multi_composed_sumcheck.rs:57-63 -> prove_internal :64-121) on two terms
 * [add, wb + wc], [mul, wb * wc].  Folds the containers in place. */
static int prove_partial_sparse(sp_t *a_add, ds_t *s_add, sp_t *a_mul, ds_t *s_mul, size_t n_vars, const fr_t *sum,
                                ora_sparse_t *round_polys, fr_t *challenges) {
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    uint8_t bytes[64 * ORA_SPARSE_MAX];
    ora_fr_to_bytes_be(bytes, sum);                              /* :70 */
    ora_transcript_commit(&tr, bytes, 32);
    fr_t xs[3];
    for (uint64_t i = 0; i < 3; ++i) ora_fr_from_u64(&xs[i], i);
    sp_t *as[2] = {a_add, a_mul};
    ds_t *ss[2] = {s_add, s_mul};
    for (size_t round = 0; round < n_vars; ++round) {
        ora_sparse_t rp;
        rp.len = 0;                                              /* :77 */
        for (int p = 0; p < 2; ++p) {
            fr_t ev[3];
            term_round_evals(ev, as[p], ss[p]);                  /* :81-90 */
            ora_sparse_t term_poly;
            ora_sparse_interpolation(&term_poly, xs, ev, 3);     /* :92-94 */
            ora_sparse_add(&rp, &rp, &term_poly);                /* :95 */
        }
        size_t nb = ora_sparse_to_bytes(bytes, &rp);             /* :98 */
        ora_transcript_commit(&tr, bytes, nb);
        ora_transcript_challenge_fr(&tr, &challenges[round]);    /* :100 */
        for (int p = 0; p < 2; ++p) {                            /* :101-107 */
            if (sp_fold(as[p], &challenges[round]) != 0) return -1;
            if (ds_fold(ss[p], &challenges[round]) != 0) return -1;
        }
        round_polys[round] = rp;
    }
    return 0;
}

/* ---- the layers -------------------------------------------------------------------------------------------------- */
static int cmp_pos(const void *a, const void *b) {
    const pos_t x = *(const pos_t *)a, y = *(const pos_t *)b;
    return x < y ? -1 : x > y;
}

/* Circuit::add_mult_mle (circuit.rs:59-97): the positions set to one, ascending, once each */
static int wiring_sparse(sp_t *add, sp_t *mul, size_t layer_index, size_t n_gates, const uint8_t *gate_type,
                         const uint32_t *in0, const uint32_t *in1) {
    const pos_t size = (pos_t)1 << (layer_index == 0 ? 3 : 3 * layer_index + 2);      /* ora_gkr_mle_size (circuit/src/utils.rs:1-10), past 64 bits */
    if (sp_alloc(add, n_gates, size) != 0 || sp_alloc(mul, n_gates, size) != 0) return -1;
    for (size_t g = 0; g < n_gates; ++g) {
        /* circuit/src/utils.rs:12-25, as gkr.c's wiring_index */
        const pos_t idx = ((pos_t)g << (2 * (layer_index + 1))) | ((pos_t)in0[g] << (layer_index + 1)) | in1[g];
        if (idx >= size) return -1;
        sp_t *t = gate_type[g] == 0 ? add : mul;
        t->pos[t->len++] = idx;
    }
    sp_t *both[2] = {add, mul};
    for (int k = 0; k < 2; ++k) {
        sp_t *t = both[k];
        qsort(t->pos, t->len, sizeof(pos_t), cmp_pos);
        size_t o = 0;
        for (size_t i = 0; i < t->len; ++i)
            if (o == 0 || t->pos[o - 1] != t->pos[i]) t->pos[o++] = t->pos[i];
        t->len = o;
        for (size_t i = 0; i < t->len; ++i) ora_fr_one(&t->val[i]);
    }
    return 0;
}

static size_t log2_exact_sz(pos_t n) {
    size_t k = 0;
    while (((pos_t)1 << k) < n) ++k;
    return k;
}

/* GKRProtocol::prove (protocol.rs:21-117), arguments as ora_gkr_prove */
int ora_gkr_prove_sparse(size_t n_layers, const size_t *n_gates, const uint8_t *gate_type, const uint32_t *in0,
                         const uint32_t *in1, const fr_t *layers, const size_t *layer_len, ora_gkr_proof_t *proof) {
    memset(proof, 0, sizeof(*proof));
    if (n_layers < 1 || n_layers > ORA_GKR_MAX_LAYERS || layer_len[0] + 1 != 2) return -1;
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    proof->w0[0] = layers[0];                                    /* :30-34 */
    ora_fr_zero(&proof->w0[1]);
    uint8_t wbytes[64];
    ora_mle_to_bytes(wbytes, proof->w0, 2);
    ora_transcript_commit(&tr, wbytes, 64);
    fr_t n_r[1];
    ora_transcript_challenge_fr(&tr, &n_r[0]);                   /* :36 */
    fr_t claimed;
    ora_mle_evaluation(&claimed, proof->w0, 2, n_r, 1);          /* :37 */

    fr_t alpha, beta;
    fr_t r_b[ORA_GKR_MAX_ROUNDS], r_c[ORA_GKR_MAX_ROUNDS], challenges[ORA_GKR_MAX_ROUNDS];
    size_t r_len = 0;
    size_t off = layer_len[0], g0 = 0;
    ora_fr_zero(&alpha);
    ora_fr_zero(&beta);
    for (size_t k = 0; k < n_layers; ++k) {                      /* k = 0: utils.rs:12-56; k >= 1: protocol.rs:61-108 */
        const fr_t *w = layers + off;
        const size_t w_len = layer_len[k + 1];
        int rc = -1;
        sp_t add = {0}, mul = {0}, a_add = {0}, a_mul = {0}, t1 = {0}, t2 = {0};
        ds_t s_add = {0}, s_mul = {0};
        if (wiring_sparse(&add, &mul, k, n_gates[k], gate_type + g0, in0 + g0, in1 + g0) != 0) goto done;
        if (k == 0) {                                            /* add_mle.partial_evaluations(&n_r, &vec![0; n_r.len()]) */
            if (sp_folds(&a_add, &add, n_r, 1) != 0 || sp_folds(&a_mul, &mul, n_r, 1) != 0) goto done;
        } else {                                                 /* (add_rb_bc * alpha) + (add_rc_bc * beta), same for mul */
            if (sp_folds(&t1, &add, r_b, r_len) != 0 || sp_folds(&t2, &add, r_c, r_len) != 0) goto done;
            if (sp_scale_add(&a_add, &t1, &alpha, &t2, &beta) != 0) goto done;
            sp_free(&t1); sp_free(&t2);
            if (sp_folds(&t1, &mul, r_b, r_len) != 0 || sp_folds(&t2, &mul, r_c, r_len) != 0) goto done;
            if (sp_scale_add(&a_mul, &t1, &alpha, &t2, &beta) != 0) goto done;
        }
        if (a_add.n != (pos_t)w_len * w_len || a_mul.n != a_add.n) goto done;   /* ComposedMultilinear::new asserts equal n_vars */
        const size_t nv = log2_exact_sz(a_add.n);
        if (((pos_t)1 << nv) != a_add.n || nv > ORA_GKR_MAX_ROUNDS) goto done;
        if (ds_init(&s_add, w, w_len, 0) != 0 || ds_init(&s_mul, w, w_len, 1) != 0) goto done;   /* wb.add_distinct(&wc), wb.mul_distinct(&wc) */
        if (prove_partial_sparse(&a_add, &s_add, &a_mul, &s_mul, nv, &claimed, proof->round_polys[k], challenges) != 0) goto done;
        proof->sums[k] = claimed;
        proof->n_rounds[k] = nv;
        memcpy(proof->challenges[k], challenges, nv * sizeof(fr_t));
        uint8_t bytes[64 * ORA_SPARSE_MAX];
        for (size_t r = 0; r < nv; ++r) {                        /* transcript.commit(&sumcheck_proof.to_bytes()) */
            size_t nb = ora_sparse_to_bytes(bytes, &proof->round_polys[k][r]);
            ora_transcript_commit(&tr, bytes, nb);
        }
        const size_t half = nv / 2;                              /* challenges.split_at(len / 2) */
        memcpy(r_b, challenges, half * sizeof(fr_t));
        memcpy(r_c, challenges + half, (nv - half) * sizeof(fr_t));
        r_len = half;
        if (ora_mle_evaluation(&proof->wb[k], w, w_len, r_b, half) != 0) goto done;
        if (ora_mle_evaluation(&proof->wc[k], w, w_len, r_c, nv - half) != 0) goto done;
        ora_transcript_challenge_fr(&tr, &alpha);
        ora_transcript_challenge_fr(&tr, &beta);
        fr_t x, y;
        ora_fr_mul(&x, &alpha, &proof->wb[k]);
        ora_fr_mul(&y, &beta, &proof->wc[k]);
        ora_fr_add(&claimed, &x, &y);
        proof->n_proofs = k + 1;
        rc = 0;
    done:
        sp_free(&add); sp_free(&mul); sp_free(&a_add); sp_free(&a_mul); sp_free(&t1); sp_free(&t2);
        if (s_add.u) ds_free(&s_add);
        if (s_mul.u) ds_free(&s_mul);
        if (rc != 0) return -1;
        off += w_len;
        g0 += n_gates[k];
    }
    return 0;
}

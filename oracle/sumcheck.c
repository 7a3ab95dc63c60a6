/*
 * sumcheck.c -- CPU ORACLE (test infrastructure): the three sumcheck provers and
 * their verifiers, restating
 *   sumcheck/src/sumcheck.rs:25-95                     (basic)
 *   sumcheck/src/composed/composed_sumcheck.rs:28-95   (product of K tables)
 *   sumcheck/src/composed/multi_composed_sumcheck.rs:36-181 (sum of products)
 * with byte encoders sumcheck/src/utils.rs:7-9,37-43,53-59.
 * The verifiers exist so that restated proofs can be self-checked (no reference
 * test pins a challenge or proof byte: see zkoracle.h "parity UNPINNED").
 */
#include "zkoracle.h"
#include <stdlib.h>
#include <string.h>

static size_t log2_floor(size_t n) {
    size_t k = 0;
    while (((size_t)2 << k) <= n) ++k;
    return k;
}

/* ---- basic ------------------------------------------------------------- */
/* sumcheck.rs:25-61 */
int ora_sumcheck_prove(const fr_t *evals, size_t n, fr_t *sum_out, fr_t *round_polys, fr_t *challenges) {
    size_t n_vars = log2_floor(n);
    if (((size_t)1 << n_vars) != n) return -1;
    ora_mle_sum(sum_out, evals, n);                       /* poly_sum :25-27 */
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    uint8_t bytes[64];
    ora_fr_to_bytes_be(bytes, sum_out);                   /* :33-35 */
    ora_transcript_commit(&tr, bytes, 32);
    fr_t *cur = (fr_t *)malloc(n * sizeof(fr_t));         /* self.poly.clone() :38 */
    fr_t *nxt = (fr_t *)malloc((n / 2 + 1) * sizeof(fr_t));
    memcpy(cur, evals, n * sizeof(fr_t));
    size_t cn = n;
    for (size_t round = 0; round < n_vars; ++round) {
        fr_t *uni = &round_polys[2 * round];
        ora_mle_half_sums(uni, cur, cn);                  /* :41 */
        ora_mle_to_bytes(bytes, uni, 2);                  /* :42 */
        ora_transcript_commit(&tr, bytes, 64);
        ora_transcript_challenge_fr(&tr, &challenges[round]); /* :46 */
        ora_mle_partial_evaluation(nxt, cur, cn, &challenges[round], 0); /* :50 */
        cn /= 2;
        fr_t *t = cur; cur = nxt; nxt = t;
    }
    free(cur); free(nxt);
    return 0;
}

/* sumcheck.rs:63-95 */
int ora_sumcheck_verify(const fr_t *evals, size_t n, const fr_t *sum, const fr_t *round_polys) {
    size_t n_vars = log2_floor(n);
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    uint8_t bytes[64];
    ora_fr_to_bytes_be(bytes, sum);
    ora_transcript_commit(&tr, bytes, 32);
    fr_t claimed = *sum;
    fr_t *challenges = (fr_t *)malloc((n_vars + 1) * sizeof(fr_t));
    int ok = 1;
    for (size_t i = 0; i < n_vars && ok; ++i) {
        const fr_t *uni = &round_polys[2 * i];
        fr_t zero, one, e0, e1, s;
        ora_fr_zero(&zero);
        ora_fr_one(&one);
        ora_mle_evaluation(&e0, uni, 2, &zero, 1);
        ora_mle_evaluation(&e1, uni, 2, &one, 1);
        ora_fr_add(&s, &e0, &e1);
        if (!ora_fr_eq(&s, &claimed)) { ok = 0; break; }
        ora_mle_to_bytes(bytes, uni, 2);
        ora_transcript_commit(&tr, bytes, 64);
        ora_transcript_challenge_fr(&tr, &challenges[i]);
        ora_mle_evaluation(&claimed, uni, 2, &challenges[i], 1);
    }
    if (ok) {
        fr_t fin;
        ora_mle_evaluation(&fin, evals, n, challenges, n_vars);
        ok = ora_fr_eq(&fin, &claimed);
    }
    free(challenges);
    return ok;
}

/* ---- composed ----------------------------------------------------------- */
/* sum over j of prod_k tables[k][j]  (composed_multilinear.rs:105-111 + composed_sumcheck.rs:28-30) */
static void product_sum(fr_t *out, const fr_t *tables, size_t k, size_t n, size_t stride) {
    ora_fr_zero(out);
    for (size_t j = 0; j < n; ++j) {
        fr_t prod = tables[j];
        for (size_t t = 1; t < k; ++t) ora_fr_mul(&prod, &prod, &tables[t * stride + j]);
        ora_fr_add(out, out, &prod);
    }
}
void ora_composed_sum(fr_t *sum, const fr_t *tables, size_t k, size_t n) { product_sum(sum, tables, k, n, n); }

/* evaluations at t = 0..=k of  sum_x prod_k fold(table_k, t)  (composed_sumcheck.rs:41-49) */
static void round_evals(fr_t *out, const fr_t *tables, size_t k, size_t n, size_t stride, fr_t *scratch) {
    for (size_t t = 0; t <= k; ++t) {
        fr_t pt;
        ora_fr_from_u64(&pt, (uint64_t)t);                       /* F::from(i as u32) */
        for (size_t q = 0; q < k; ++q)
            ora_mle_partial_evaluation(scratch + q * (n / 2), tables + q * stride, n, &pt, 0);
        product_sum(&out[t], scratch, k, n / 2, n / 2);
    }
}

/* composed_sumcheck.rs:32-67 */
int ora_composed_prove(const fr_t *tables, size_t k, size_t n, fr_t *round_polys, fr_t *challenges) {
    size_t n_vars = log2_floor(n);
    if (((size_t)1 << n_vars) != n) return -1;
    ora_transcript_t tr;
    ora_transcript_new(&tr);                                   /* nothing absorbed first :33 */
    fr_t *cur = (fr_t *)malloc(k * n * sizeof(fr_t));
    fr_t *nxt = (fr_t *)malloc(k * (n / 2 + 1) * sizeof(fr_t));
    fr_t *scratch = (fr_t *)malloc(k * (n / 2 + 1) * sizeof(fr_t));
    uint8_t *bytes = (uint8_t *)malloc(32 * (k + 1));
    memcpy(cur, tables, k * n * sizeof(fr_t));
    size_t cn = n;
    for (size_t round = 0; round < n_vars; ++round) {
        fr_t *rp = &round_polys[(k + 1) * round];
        round_evals(rp, cur, k, cn, cn, scratch);
        ora_mle_to_bytes(bytes, rp, k + 1);                    /* vec_to_bytes :51 */
        ora_transcript_commit(&tr, bytes, 32 * (k + 1));
        ora_transcript_challenge_fr(&tr, &challenges[round]);
        for (size_t q = 0; q < k; ++q)                         /* :57 */
            ora_mle_partial_evaluation(nxt + q * (cn / 2), cur + q * cn, cn, &challenges[round], 0);
        cn /= 2;
        memcpy(cur, nxt, k * cn * sizeof(fr_t));
    }
    free(cur); free(nxt); free(scratch); free(bytes);
    return 0;
}

/* composed_sumcheck.rs:69-95 */
int ora_composed_verify(const fr_t *tables, size_t k, size_t n, const fr_t *sum, const fr_t *round_polys) {
    size_t n_vars = log2_floor(n);
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    fr_t claimed = *sum;
    fr_t *challenges = (fr_t *)malloc((n_vars + 1) * sizeof(fr_t));
    uint8_t *bytes = (uint8_t *)malloc(32 * (k + 1));
    int ok = 1;
    for (size_t round = 0; round < n_vars; ++round) {
        const fr_t *rp = &round_polys[(k + 1) * round];
        ora_mle_to_bytes(bytes, rp, k + 1);
        ora_transcript_commit(&tr, bytes, 32 * (k + 1));
        ora_transcript_challenge_fr(&tr, &challenges[round]);
        fr_t xs[ORA_SPARSE_MAX];
        for (size_t i = 0; i <= k; ++i) ora_fr_from_u64(&xs[i], (uint64_t)i);
        ora_sparse_t uni;
        ora_sparse_interpolation(&uni, xs, rp, k + 1);
        fr_t zero, one, e0, e1, s;
        ora_fr_zero(&zero);
        ora_fr_one(&one);
        ora_sparse_evaluate(&e0, &uni, &zero);
        ora_sparse_evaluate(&e1, &uni, &one);
        ora_fr_add(&s, &e0, &e1);
        if (!ora_fr_eq(&s, &claimed)) { ok = 0; break; }
        ora_sparse_evaluate(&claimed, &uni, &challenges[round]);
    }
    if (ok) {
        fr_t prod, e;
        ora_fr_one(&prod);                                       /* composed_multilinear.rs:52-61 */
        for (size_t q = 0; q < k; ++q) {
            ora_mle_evaluation(&e, tables + q * n, n, challenges, n_vars);
            ora_fr_mul(&prod, &prod, &e);
        }
        ok = ora_fr_eq(&prod, &claimed);
    }
    free(challenges); free(bytes);
    return ok;
}

/* ---- multi-composed ------------------------------------------------------ */
/* multi_composed_sumcheck.rs:36-45 */
void ora_multi_composed_sum(fr_t *sum, const fr_t *tables, const size_t *term_sizes, size_t n_terms, size_t n) {
    ora_fr_zero(sum);
    size_t off = 0;
    for (size_t p = 0; p < n_terms; ++p) {
        fr_t s;
        product_sum(&s, tables + off * n, term_sizes[p], n, n);
        ora_fr_add(sum, sum, &s);
        off += term_sizes[p];
    }
}

/* multi_composed_sumcheck.rs:47-121 (prove: partial == 0; prove_partial: partial != 0) */
int ora_multi_composed_prove(const fr_t *tables, const size_t *term_sizes, size_t n_terms, size_t n,
                             const fr_t *sum, int partial, ora_sparse_t *round_polys, fr_t *challenges) {
    size_t n_vars = log2_floor(n);
    if (((size_t)1 << n_vars) != n) return -1;
    size_t total = 0, kmax = 0;
    for (size_t p = 0; p < n_terms; ++p) {
        total += term_sizes[p];
        if (term_sizes[p] > kmax) kmax = term_sizes[p];
    }
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    uint8_t bytes[64 * ORA_SPARSE_MAX];
    if (!partial) {                                             /* :51-53 composed_poly_to_bytes of every table */
        uint8_t *tb = (uint8_t *)malloc(32 * n);
        for (size_t q = 0; q < total; ++q) {
            ora_mle_to_bytes(tb, tables + q * n, n);
            ora_transcript_commit(&tr, tb, 32 * n);
        }
        free(tb);
    }
    ora_fr_to_bytes_be(bytes, sum);                             /* :70 */
    ora_transcript_commit(&tr, bytes, 32);
    fr_t *cur = (fr_t *)malloc(total * n * sizeof(fr_t));
    fr_t *nxt = (fr_t *)malloc(total * (n / 2 + 1) * sizeof(fr_t));
    fr_t *scratch = (fr_t *)malloc(kmax * (n / 2 + 1) * sizeof(fr_t));
    memcpy(cur, tables, total * n * sizeof(fr_t));
    size_t cn = n;
    for (size_t round = 0; round < n_vars; ++round) {
        ora_sparse_t rp;
        rp.len = 0;                                              /* SparseUnivariatePolynomial::zero() :77 */
        size_t off = 0;
        for (size_t p = 0; p < n_terms; ++p) {
            size_t k = term_sizes[p];
            fr_t ev[ORA_SPARSE_MAX], xs[ORA_SPARSE_MAX];
            round_evals(ev, cur + off * cn, k, cn, cn, scratch); /* :81-90 */
            for (size_t i = 0; i <= k; ++i) ora_fr_from_u64(&xs[i], (uint64_t)i);
            ora_sparse_t term_poly;
            ora_sparse_interpolation(&term_poly, xs, ev, k + 1); /* :92-94 */
            ora_sparse_add(&rp, &rp, &term_poly);                /* :95 */
            off += k;
        }
        size_t nb = ora_sparse_to_bytes(bytes, &rp);             /* :98 */
        ora_transcript_commit(&tr, bytes, nb);
        ora_transcript_challenge_fr(&tr, &challenges[round]);
        for (size_t q = 0; q < total; ++q)                       /* :101-107 */
            ora_mle_partial_evaluation(nxt + q * (cn / 2), cur + q * cn, cn, &challenges[round], 0);
        cn /= 2;
        memcpy(cur, nxt, total * cn * sizeof(fr_t));
        round_polys[round] = rp;
    }
    free(cur); free(nxt); free(scratch);
    return 0;
}

/* multi_composed_sumcheck.rs:126-181 (verify = full: table bytes absorbed first + oracle check) */
int ora_multi_composed_verify(const fr_t *tables, const size_t *term_sizes, size_t n_terms, size_t n,
                              const fr_t *sum, const ora_sparse_t *round_polys, size_t n_rounds) {
    size_t n_vars = log2_floor(n);
    size_t total = 0;
    for (size_t p = 0; p < n_terms; ++p) total += term_sizes[p];
    ora_transcript_t tr;
    ora_transcript_new(&tr);
    uint8_t *tb = (uint8_t *)malloc(32 * n);
    for (size_t q = 0; q < total; ++q) {
        ora_mle_to_bytes(tb, tables + q * n, n);
        ora_transcript_commit(&tr, tb, 32 * n);
    }
    free(tb);
    uint8_t bytes[64 * ORA_SPARSE_MAX];
    ora_fr_to_bytes_be(bytes, sum);
    ora_transcript_commit(&tr, bytes, 32);
    fr_t claimed = *sum;
    fr_t *challenges = (fr_t *)malloc((n_rounds + 1) * sizeof(fr_t));
    for (size_t round = 0; round < n_rounds; ++round) {
        const ora_sparse_t *rp = &round_polys[round];
        size_t nb = ora_sparse_to_bytes(bytes, rp);
        ora_transcript_commit(&tr, bytes, nb);
        ora_transcript_challenge_fr(&tr, &challenges[round]);
        fr_t zero, one, e0, e1, s;
        ora_fr_zero(&zero);
        ora_fr_one(&one);
        ora_sparse_evaluate(&e0, rp, &zero);
        ora_sparse_evaluate(&e1, rp, &one);
        ora_fr_add(&s, &e0, &e1);
        if (!ora_fr_eq(&s, &claimed)) { free(challenges); return -1; }   /* Err("Verification failed") :170 */
        ora_sparse_evaluate(&claimed, rp, &challenges[round]);
    }
    fr_t acc;
    ora_fr_zero(&acc);
    size_t off = 0;
    for (size_t p = 0; p < n_terms; ++p) {                       /* :134-139 oracle check */
        fr_t prod, e;
        ora_fr_one(&prod);
        for (size_t q = 0; q < term_sizes[p]; ++q) {
            ora_mle_evaluation(&e, tables + (off + q) * n, n, challenges, n_vars);
            ora_fr_mul(&prod, &prod, &e);
        }
        ora_fr_add(&acc, &acc, &prod);
        off += term_sizes[p];
    }
    free(challenges);
    return ora_fr_eq(&acc, &claimed);
}

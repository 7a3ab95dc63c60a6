/*
 * multilinear.c -- CPU ORACLE (test infrastructure): evaluation-form multilinear
 * polynomial, restating polynomial/src/multilinear/evaluation_form.rs and
 * pick_pairs_with_random_index (polynomial/src/utils.rs:26-53).
 *
 * Index convention (evaluation_form.rs:328-359 KATs): variable k <-> index bit
 * (n_vars-1-k); variable 0 is the most significant bit.
 */
#include "zkoracle.h"
#include <stdlib.h>
#include <string.h>

/* utils.rs:26-53.  The reference materialises Vec<(usize,usize)>; the pairs it
 * yields, in order, are: for block b in 0..2^k, for y in 0..half:
 * (b*2*half + y, b*2*half + half + y) with half = n / 2^k / 2.  We walk the same
 * sequence without allocating.  evaluation_form.rs:123-141: out.push(r*y2 + (1-r)*y1). */
int ora_mle_partial_evaluation(fr_t *out, const fr_t *in, size_t n, const fr_t *r, size_t var_index) {
    if (n % 2 != 0) return -1;             /* assert!(n % 2 == 0) utils.rs:30 */
    if (!(var_index < n / 2)) return -2;   /* assert!(variable_index < n/2) utils.rs:31-34 */
    size_t iters = (size_t)1 << var_index;
    size_t half = (n / iters) / 2;
    fr_t one, one_minus_r;
    ora_fr_one(&one);
    for (size_t b = 0; b < iters; ++b) {
        for (size_t y = 0; y < half; ++y) {
            size_t i = b * 2 * half + y;
            size_t j = i + half;
            fr_t t1, t2;
            ora_fr_sub(&one_minus_r, &one, r);      /* (F::one() - eval_point), recomputed per pair as in :133 */
            ora_fr_mul(&t1, r, &in[j]);
            ora_fr_mul(&t2, &one_minus_r, &in[i]);
            ora_fr_add(&out[b * half + y], &t1, &t2);
        }
    }
    return 0;
}

int ora_mle_partial_evaluation_mt(fr_t *out, const fr_t *in, size_t n, const fr_t *r, size_t var_index) {
    if (n % 2 != 0) return -1;
    if (!(var_index < n / 2)) return -2;
    size_t iters = (size_t)1 << var_index;
    size_t half = (n / iters) / 2;
    fr_t one, one_minus_r;
    ora_fr_one(&one);
    ora_fr_sub(&one_minus_r, &one, r);
#pragma omp parallel for schedule(static)
    for (size_t q = 0; q < n / 2; ++q) {
        size_t b = q / half, y = q % half;
        size_t i = b * 2 * half + y, j = i + half;
        fr_t t1, t2;
        ora_fr_mul(&t1, r, &in[j]);
        ora_fr_mul(&t2, &one_minus_r, &in[i]);
        ora_fr_add(&out[q], &t1, &t2);
    }
    return 0;
}

/* evaluation_form.rs:143-159 */
int ora_mle_partial_evaluations(fr_t *out, size_t *out_n, const fr_t *in, size_t n, const fr_t *pts,
                                const size_t *var_indices, size_t n_pts) {
    fr_t *cur = (fr_t *)malloc(n * sizeof(fr_t));   /* self.clone() :144 */
    fr_t *nxt = (fr_t *)malloc((n / 2 + 1) * sizeof(fr_t));
    memcpy(cur, in, n * sizeof(fr_t));
    size_t cn = n;
    int rc = 0;
    for (size_t i = 0; i < n_pts; ++i) {
        rc = ora_mle_partial_evaluation(nxt, cur, cn, &pts[i], var_indices[i]);
        if (rc) break;
        cn /= 2;
        memcpy(cur, nxt, cn * sizeof(fr_t));
    }
    if (!rc) { memcpy(out, cur, cn * sizeof(fr_t)); *out_n = cn; }
    free(cur); free(nxt);
    return rc;
}

/* evaluation_form.rs:162-175 : n_vars successive folds of variable 0 */
int ora_mle_evaluation(fr_t *out, const fr_t *in, size_t n, const fr_t *pts, size_t n_pts) {
    size_t n_vars = 0;
    while (((size_t)1 << n_vars) < n) ++n_vars;
    if (n_pts != n_vars) return -3;   /* assert_eq! :163-167 */
    if (n == 1) { *out = in[0]; return 0; }
    fr_t *cur = (fr_t *)malloc(n * sizeof(fr_t));
    fr_t *nxt = (fr_t *)malloc((n / 2) * sizeof(fr_t));
    memcpy(cur, in, n * sizeof(fr_t));
    size_t cn = n;
    for (size_t i = 0; i < n_pts; ++i) {
        ora_mle_partial_evaluation(nxt, cur, cn, &pts[i], 0);
        cn /= 2;
        fr_t *t = cur; cur = nxt; nxt = t;
    }
    *out = cur[0];
    free(cur); free(nxt);
    return 0;
}

/* evaluation_form.rs:68-74 */
void ora_mle_half_sums(fr_t out[2], const fr_t *in, size_t n) {
    size_t mid = n / 2;
    ora_fr_zero(&out[0]);
    ora_fr_zero(&out[1]);
    for (size_t i = 0; i < mid; ++i) ora_fr_add(&out[0], &out[0], &in[i]);
    for (size_t i = mid; i < n; ++i) ora_fr_add(&out[1], &out[1], &in[i]);
}
/* evaluation_form.rs:80-84 ; sumcheck.rs:25-27 */
void ora_mle_sum(fr_t *out, const fr_t *in, size_t n) {
    ora_fr_zero(out);
    for (size_t i = 0; i < n; ++i) ora_fr_add(out, out, &in[i]);
}
/* evaluation_form.rs:28-39 */
void ora_mle_add_distinct(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    for (size_t i = 0; i < na; ++i)
        for (size_t j = 0; j < nb; ++j) ora_fr_add(&out[i * nb + j], &a[i], &b[j]);
}
/* evaluation_form.rs:41-52 */
void ora_mle_mul_distinct(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    for (size_t i = 0; i < na; ++i)
        for (size_t j = 0; j < nb; ++j) ora_fr_mul(&out[i * nb + j], &a[i], &b[j]);
}
/* evaluation_form.rs:54-62 */
void ora_mle_to_bytes(uint8_t *out, const fr_t *in, size_t n) {
    for (size_t i = 0; i < n; ++i) ora_fr_to_bytes_be(out + 32 * i, &in[i]);
}

/* add_to_front (evaluation_form.rs:86-96): the table followed by itself, 2^variable_length times over:
 * out has n * 2 * 2^variable_length entries. */
void ora_mle_add_to_front(fr_t *out, const fr_t *in, size_t n, size_t variable_length) {
    const size_t copies = (size_t)2 << variable_length;
    for (size_t r = 0; r < copies; ++r) memcpy(out + r * n, in, n * sizeof(fr_t));
}
/* add_to_back (evaluation_form.rs:98-110): every entry repeated 2^variable_length times in place */
void ora_mle_add_to_back(fr_t *out, const fr_t *in, size_t n, size_t variable_length) {
    const size_t reps = (size_t)1 << variable_length;
    for (size_t i = 0; i < n; ++i)
        for (size_t r = 0; r < reps; ++r) out[i * reps + r] = in[i];
}

/*
 * sparse.c -- CPU ORACLE (test infrastructure): SparseUnivariatePolynomial as
 * used for multi-composed sumcheck round polynomials, restating
 * polynomial/src/univariate/sparse_univariate.rs:27-34,40-63,90-106,159-203 and
 * lagrange_basis (polynomial/src/utils.rs:78-100).
 */
#include "zkoracle.h"
#include <string.h>

/* utils.rs:78-100 : coefficients (low degree first) of the i-th Lagrange basis polynomial */
static void lagrange_basis(fr_t *l_i, size_t *len, const fr_t *xs, size_t n, size_t i) {
    fr_t cur[ORA_SPARSE_MAX + 1], nxt[ORA_SPARSE_MAX + 1];
    size_t cl = 1;
    ora_fr_one(&cur[0]);
    for (size_t j = 0; j < n; ++j) {
        if (j == i) continue;
        for (size_t k = 0; k <= cl; ++k) ora_fr_zero(&nxt[k]);
        for (size_t k = 0; k < cl; ++k) {
            fr_t t;
            ora_fr_mul(&t, &cur[k], &xs[j]);
            ora_fr_sub(&nxt[k], &nxt[k], &t);          /* new_l_i[k] -= coeff * x_j */
            ora_fr_add(&nxt[k + 1], &nxt[k + 1], &cur[k]); /* new_l_i[k+1] += coeff */
        }
        cl += 1;
        memcpy(cur, nxt, cl * sizeof(fr_t));
    }
    fr_t denom, dinv;
    ora_fr_one(&denom);
    for (size_t j = 0; j < n; ++j) {
        if (j == i) continue;
        fr_t d;
        ora_fr_sub(&d, &xs[i], &xs[j]);
        ora_fr_mul(&denom, &denom, &d);
    }
    ora_fr_inv(&dinv, &denom);
    for (size_t k = 0; k < cl; ++k) ora_fr_mul(&l_i[k], &cur[k], &dinv);
    *len = cl;
}

/* sparse_univariate.rs:40-63 : zero coefficients are dropped HERE (filter at :55) */
int ora_sparse_interpolation(ora_sparse_t *o, const fr_t *xs, const fr_t *ys, size_t n) {
    if (n > ORA_SPARSE_MAX) return -1;
    fr_t result[ORA_SPARSE_MAX];
    for (size_t k = 0; k < n; ++k) ora_fr_zero(&result[k]);
    for (size_t i = 0; i < n; ++i) {
        fr_t l_i[ORA_SPARSE_MAX + 1];
        size_t len;
        lagrange_basis(l_i, &len, xs, n, i);
        for (size_t k = 0; k < len; ++k) {
            fr_t t;
            ora_fr_mul(&t, &l_i[k], &ys[i]);
            ora_fr_add(&result[k], &result[k], &t);
        }
    }
    o->len = 0;
    for (size_t k = 0; k < n; ++k) {
        if (ora_fr_is_zero(&result[k])) continue;
        o->coeff[o->len] = result[k];
        ora_fr_from_u64(&o->pow[o->len], (uint64_t)k);
        o->len++;
    }
    return 0;
}

/* Ord on ark-ff Fp compares into_bigint() values */
static int fr_cmp(const fr_t *a, const fr_t *b) {
    uint64_t ca[4], cb[4];
    ora_fr_to_canonical(ca, a);
    ora_fr_to_canonical(cb, b);
    for (int i = 3; i >= 0; --i) {
        if (ca[i] < cb[i]) return -1;
        if (ca[i] > cb[i]) return 1;
    }
    return 0;
}

/* sparse_univariate.rs:159-203 : ordered merge; equal powers are summed and KEPT even if the sum is zero */
void ora_sparse_add(ora_sparse_t *o, const ora_sparse_t *a, const ora_sparse_t *b) {
    ora_sparse_t r;
    r.len = 0;
    size_t i = 0, j = 0;
    while (i < a->len || j < b->len) {
        if (i < a->len && j < b->len) {
            int c = fr_cmp(&a->pow[i], &b->pow[j]);
            if (c == 0) {
                ora_fr_add(&r.coeff[r.len], &a->coeff[i], &b->coeff[j]);
                r.pow[r.len] = a->pow[i];
                ++i; ++j;
            } else if (c < 0) {
                r.coeff[r.len] = a->coeff[i]; r.pow[r.len] = a->pow[i]; ++i;
            } else {
                r.coeff[r.len] = b->coeff[j]; r.pow[r.len] = b->pow[j]; ++j;
            }
        } else if (i < a->len) {
            r.coeff[r.len] = a->coeff[i]; r.pow[r.len] = a->pow[i]; ++i;
        } else {
            r.coeff[r.len] = b->coeff[j]; r.pow[r.len] = b->pow[j]; ++j;
        }
        r.len++;
    }
    *o = r;
}

/* sparse_univariate.rs:90-106 : sum coeff * point.pow(pow.into_bigint()) */
void ora_sparse_evaluate(fr_t *o, const ora_sparse_t *p, const fr_t *x) {
    fr_t acc;
    ora_fr_zero(&acc);
    for (size_t k = 0; k < p->len; ++k) {
        uint64_t e[4];
        ora_fr_to_canonical(e, &p->pow[k]);
        fr_t pw, t;
        ora_fr_one(&pw);
        for (int bit = 255; bit >= 0; --bit) {
            ora_fr_mul(&pw, &pw, &pw);
            if ((e[bit / 64] >> (bit % 64)) & 1) ora_fr_mul(&pw, &pw, x);
        }
        ora_fr_mul(&t, &p->coeff[k], &pw);
        ora_fr_add(&acc, &acc, &t);
    }
    *o = acc;
}

/* sparse_univariate.rs:27-34 : coeff || pow, 32 bytes big-endian canonical each */
size_t ora_sparse_to_bytes(uint8_t *out, const ora_sparse_t *p) {
    for (size_t k = 0; k < p->len; ++k) {
        ora_fr_to_bytes_be(out + 64 * k, &p->coeff[k]);
        ora_fr_to_bytes_be(out + 64 * k + 32, &p->pow[k]);
    }
    return 64 * p->len;
}

/*
 * ntt.c -- CPU ORACLE (test infrastructure): radix-2 NTT over BLS12-381 Fr and
 * the polynomial product built on it, restating
 *   serial_fft / bitreverse      polynomial/src/utils.rs:281-324
 *   Domain::{new,fft,ifft,...}   polynomial/src/univariate/domain.rs:31-48,108-133
 *   UnivariateEval::multiply     polynomial/src/univariate/evaluation.rs:59-86
 *   DenseUnivariatePolynomial Mul / evaluate  dense_univariate.rs:184-196,210-233
 */
#include "zkoracle.h"
#include <stdlib.h>
#include <string.h>

/* utils.rs:317-324 */
static uint32_t bitreverse(uint32_t n, uint32_t l) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < l; ++i) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}

/* utils.rs:281-315 : in-place bit-reversal then size_log DIT butterfly stages */
void ora_serial_fft(fr_t *list, size_t n_, const fr_t *w, uint32_t size_log) {
    uint32_t n = (uint32_t)n_;
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t rk = bitreverse(k, size_log);
        if (k < rk) { fr_t t = list[rk]; list[rk] = list[k]; list[k] = t; }
    }
    uint32_t m = 1;
    for (uint32_t s = 0; s < size_log; ++s) {
        fr_t w_m;
        ora_fr_pow_u64(&w_m, w, (uint64_t)(n / (2 * m)));   /* w.pow([n/(2m)]) :295 */
        for (uint32_t k = 0; k < n; k += 2 * m) {
            fr_t wj;
            ora_fr_one(&wj);
            for (uint32_t j = 0; j < m; ++j) {
                fr_t t, tmp;
                ora_fr_mul(&t, &list[k + j + m], &wj);
                ora_fr_sub(&tmp, &list[k + j], &t);
                list[k + j + m] = tmp;
                ora_fr_add(&list[k + j], &list[k + j], &t);
                ora_fr_mul(&wj, &wj, &w_m);
            }
        }
        m *= 2;
    }
}

static uint32_t log2_exact(size_t n) {
    uint32_t k = 0;
    while (((size_t)1 << k) < n) ++k;
    return k;
}

/* domain.rs:108-123 : resize to domain size with zeros, forward transform with omega */
int ora_domain_fft(fr_t *out, const fr_t *coeffs, size_t n_coeffs, size_t domain_size) {
    fr_t w;
    if (n_coeffs > domain_size || !ora_fr_get_root_of_unity(&w, (uint64_t)domain_size)) return -1;
    memcpy(out, coeffs, n_coeffs * sizeof(fr_t));
    for (size_t i = n_coeffs; i < domain_size; ++i) ora_fr_zero(&out[i]);
    ora_serial_fft(out, domain_size, &w, log2_exact(domain_size));
    return 0;
}
/* domain.rs:114-133 : transform with omega^-1 then scale by size^-1 */
int ora_domain_ifft(fr_t *out, const fr_t *evals, size_t n_evals, size_t domain_size) {
    fr_t w, winv, nf, ninv;
    if (n_evals > domain_size || !ora_fr_get_root_of_unity(&w, (uint64_t)domain_size)) return -1;
    ora_fr_inv(&winv, &w);
    ora_fr_from_u64(&nf, (uint64_t)domain_size);
    ora_fr_inv(&ninv, &nf);
    memcpy(out, evals, n_evals * sizeof(fr_t));
    for (size_t i = n_evals; i < domain_size; ++i) ora_fr_zero(&out[i]);
    ora_serial_fft(out, domain_size, &winv, log2_exact(domain_size));
    for (size_t i = 0; i < domain_size; ++i) ora_fr_mul(&out[i], &out[i], &ninv);
    return 0;
}

/* evaluation.rs:59-86 */
int ora_univariate_multiply(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    if (na == 0 || nb == 0) return -1;      /* len_a + len_b - 1 underflows in the reference */
    size_t unscaled = na + nb - 1;
    size_t len = 1;
    while (len < unscaled) len <<= 1;
    fr_t *ea = (fr_t *)malloc(len * sizeof(fr_t));
    fr_t *eb = (fr_t *)malloc(len * sizeof(fr_t));
    fr_t *res = (fr_t *)malloc(len * sizeof(fr_t));
    int rc = ora_domain_fft(ea, a, na, len);
    if (!rc) rc = ora_domain_fft(eb, b, nb, len);
    if (!rc) {
        for (size_t i = 0; i < len; ++i) ora_fr_mul(&ea[i], &ea[i], &eb[i]);
        rc = ora_domain_ifft(res, ea, len, len);
    }
    if (!rc) memcpy(out, res, unscaled * sizeof(fr_t));
    free(ea); free(eb); free(res);
    return rc;
}

static size_t dense_degree(const fr_t *a, size_t n) {   /* dense_univariate.rs:199-207 */
    while (n > 0 && ora_fr_is_zero(&a[n - 1])) --n;
    return n == 0 ? 0 : n - 1;
}
/* dense_univariate.rs:210-233 */
size_t ora_dense_mul(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    if (na == 0 || nb == 0) return 0;
    size_t da = dense_degree(a, na), db = dense_degree(b, nb);
    for (size_t i = 0; i <= da + db; ++i) ora_fr_zero(&out[i]);
    for (size_t i = 0; i <= da; ++i)
        for (size_t j = 0; j <= db; ++j) {
            fr_t t;
            ora_fr_mul(&t, &a[i], &b[j]);
            ora_fr_add(&out[i + j], &out[i + j], &t);
        }
    return da + db + 1;
}
/* dense_univariate.rs:184-196 */
void ora_dense_evaluate(fr_t *o, const fr_t *coeffs, size_t n, const fr_t *x) {
    fr_t acc;
    ora_fr_zero(&acc);
    for (size_t i = 0; i < n; ++i) {
        fr_t pw, t;
        ora_fr_pow_u64(&pw, x, (uint64_t)i);
        ora_fr_mul(&t, &coeffs[i], &pw);
        ora_fr_add(&acc, &acc, &t);
    }
    *o = acc;
}

/* divide_with_q_and_r (dense_univariate.rs:88-124): quotient and remainder of a / b, coefficient vectors low degree
 * first; `degree` ignores zero leading coefficients (:199-207), `is_zero` means an EMPTY vector (:41-43).
 * q must hold na entries, r na entries; returns 0 and the lengths, -1 for "Dividing by zero polynomial". */
int ora_dense_divide(fr_t *q, size_t *nq, fr_t *r, size_t *nr, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    if (na == 0) { *nq = 0; *nr = 0; return 0; }
    if (nb == 0) return -1;
    const size_t da = dense_degree(a, na), db = dense_degree(b, nb);
    if (da < db) { *nq = 0; memcpy(r, a, na * sizeof(fr_t)); *nr = na; return 0; }
    const size_t lq = da - db + 1;
    for (size_t i = 0; i < lq; ++i) ora_fr_zero(&q[i]);
    memcpy(r, a, na * sizeof(fr_t));
    size_t lr = na;
    fr_t inv;
    if (!ora_fr_inv(&inv, &b[nb - 1])) return -1;                 /* leading_coefficient().inverse().unwrap() */
    while (lr != 0 && dense_degree(r, lr) >= db) {
        fr_t cur;
        ora_fr_mul(&cur, &r[lr - 1], &inv);                        /* remainder.coefficients.last() * divisor_leading_inv */
        const size_t cur_deg = dense_degree(r, lr) - db;
        q[cur_deg] = cur;
        for (size_t i = 0; i < nb; ++i) {
            fr_t t;
            ora_fr_mul(&t, &cur, &b[i]);
            ora_fr_sub(&r[cur_deg + i], &r[cur_deg + i], &t);
        }
        while (lr != 0 && ora_fr_is_zero(&r[lr - 1])) --lr;       /* pop zero leading coefficients */
    }
    *nq = lq;
    *nr = lr;
    return 0;
}

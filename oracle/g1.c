/*
 * g1.c -- CPU ORACLE (test infrastructure): BLS12-381 G1 (y^2 = x^3 + 4) group
 * law and the KZG SRS / commitment of the reference, restating
 *   kzg/src/trusted_setup.rs:25-35, kzg/src/utils.rs:19-40,
 *   polynomial/src/utils.rs:141-157 (boolean_hypercube, MSB first),
 *   kzg/src/univariate_kzg.rs:18-58, kzg/src/multilinear_kzg.rs:33-48.
 * The group law itself lives in ark-ec ^0.4.2 (`short_weierstrass::Projective`,
 * Jacobian coordinates; not vendored in the reference).  Jacobian X,Y,Z are
 * algorithm-dependent, so parity is defined on the AFFINE result only; the
 * formulas here are the standard a=0 Jacobian add-2007-bl / dbl-2009-l.
 */
#include "zkoracle.h"
#include <stdlib.h>
#include <string.h>

static const uint64_t GX_CANON[6] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                                     0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
static const uint64_t GY_CANON[6] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                                     0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};

static int fq_is_zero(const fq_t *a) {
    uint64_t x = 0;
    for (int i = 0; i < 6; ++i) x |= a->l[i];
    return x == 0;
}
static int fq_eq(const fq_t *a, const fq_t *b) { return memcmp(a->l, b->l, 48) == 0; }
static void fq_zero(fq_t *a) { memset(a->l, 0, 48); }
static void fq_one(fq_t *a) {
    uint64_t one[6] = {1, 0, 0, 0, 0, 0};
    ora_fq_from_canonical(a, one);
}
static void fq_dbl(fq_t *o, const fq_t *a) { ora_fq_add(o, a, a); }

void ora_g1_generator(g1_jac_t *o) {
    ora_fq_from_canonical(&o->x, GX_CANON);
    ora_fq_from_canonical(&o->y, GY_CANON);
    fq_one(&o->z);
}
void ora_g1_identity(g1_jac_t *o) {   /* P::G1::default(): Z = 0 */
    fq_one(&o->x);
    fq_one(&o->y);
    fq_zero(&o->z);
}
void ora_g1_neg(g1_jac_t *o, const g1_jac_t *a) {
    fq_t z;
    fq_zero(&z);
    o->x = a->x;
    ora_fq_sub(&o->y, &z, &a->y);
    o->z = a->z;
}

/* dbl-2009-l, a = 0 */
void ora_g1_double(g1_jac_t *o, const g1_jac_t *p) {
    if (fq_is_zero(&p->z)) { *o = *p; return; }
    fq_t a, b, c, d, e, f, t, x3, y3, z3;
    ora_fq_mul(&a, &p->x, &p->x);
    ora_fq_mul(&b, &p->y, &p->y);
    ora_fq_mul(&c, &b, &b);
    ora_fq_add(&t, &p->x, &b);
    ora_fq_mul(&t, &t, &t);
    ora_fq_sub(&t, &t, &a);
    ora_fq_sub(&t, &t, &c);
    fq_dbl(&d, &t);
    fq_dbl(&e, &a);
    ora_fq_add(&e, &e, &a);
    ora_fq_mul(&f, &e, &e);
    fq_dbl(&t, &d);
    ora_fq_sub(&x3, &f, &t);
    ora_fq_sub(&t, &d, &x3);
    ora_fq_mul(&y3, &e, &t);
    fq_dbl(&t, &c); fq_dbl(&t, &t); fq_dbl(&t, &t);
    ora_fq_sub(&y3, &y3, &t);
    ora_fq_mul(&z3, &p->y, &p->z);
    fq_dbl(&z3, &z3);
    o->x = x3; o->y = y3; o->z = z3;
}

/* add-2007-bl with identity / doubling / inverse cases */
void ora_g1_add(g1_jac_t *o, const g1_jac_t *p, const g1_jac_t *q) {
    if (fq_is_zero(&p->z)) { *o = *q; return; }
    if (fq_is_zero(&q->z)) { *o = *p; return; }
    fq_t z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    ora_fq_mul(&z1z1, &p->z, &p->z);
    ora_fq_mul(&z2z2, &q->z, &q->z);
    ora_fq_mul(&u1, &p->x, &z2z2);
    ora_fq_mul(&u2, &q->x, &z1z1);
    ora_fq_mul(&s1, &p->y, &q->z);
    ora_fq_mul(&s1, &s1, &z2z2);
    ora_fq_mul(&s2, &q->y, &p->z);
    ora_fq_mul(&s2, &s2, &z1z1);
    if (fq_eq(&u1, &u2)) {
        if (fq_eq(&s1, &s2)) { ora_g1_double(o, p); return; }
        ora_g1_identity(o);
        return;
    }
    ora_fq_sub(&h, &u2, &u1);
    fq_dbl(&i, &h);
    ora_fq_mul(&i, &i, &i);
    ora_fq_mul(&j, &h, &i);
    ora_fq_sub(&r, &s2, &s1);
    fq_dbl(&r, &r);
    ora_fq_mul(&v, &u1, &i);
    ora_fq_mul(&x3, &r, &r);
    ora_fq_sub(&x3, &x3, &j);
    fq_dbl(&t, &v);
    ora_fq_sub(&x3, &x3, &t);
    ora_fq_sub(&t, &v, &x3);
    ora_fq_mul(&y3, &r, &t);
    ora_fq_mul(&t, &s1, &j);
    fq_dbl(&t, &t);
    ora_fq_sub(&y3, &y3, &t);
    ora_fq_add(&z3, &p->z, &q->z);
    ora_fq_mul(&z3, &z3, &z3);
    ora_fq_sub(&z3, &z3, &z1z1);
    ora_fq_sub(&z3, &z3, &z2z2);
    ora_fq_mul(&z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}

/* Group::mul_bigint (ark-ec): double-and-add over the scalar bits, MSB first.
 * Call sites: univariate_kzg.rs:27,53; multilinear_kzg.rs:46; trusted_setup.rs:33. */
void ora_g1_mul_bigint(g1_jac_t *o, const g1_jac_t *base, const uint64_t *scalar, size_t n_limbs) {
    g1_jac_t res;
    ora_g1_identity(&res);
    int started = 0;
    for (int i = (int)n_limbs * 64 - 1; i >= 0; --i) {
        int bit = (int)((scalar[i / 64] >> (i % 64)) & 1);
        if (started) ora_g1_double(&res, &res);
        if (bit) { ora_g1_add(&res, &res, base); started = 1; }
    }
    *o = res;
}

void ora_g1_to_affine(g1_affine_t *o, const g1_jac_t *a) {
    if (fq_is_zero(&a->z)) {
        fq_zero(&o->x); fq_zero(&o->y); o->inf = 1;
        return;
    }
    fq_t zi, zi2, zi3;
    ora_fq_inv(&zi, &a->z);
    ora_fq_mul(&zi2, &zi, &zi);
    ora_fq_mul(&zi3, &zi2, &zi);
    ora_fq_mul(&o->x, &a->x, &zi2);
    ora_fq_mul(&o->y, &a->y, &zi3);
    o->inf = 0;
}
void ora_g1_from_affine(g1_jac_t *o, const g1_affine_t *a) {
    if (a->inf) { ora_g1_identity(o); return; }
    o->x = a->x; o->y = a->y;
    fq_one(&o->z);
}
int ora_g1_is_on_curve(const g1_affine_t *a) {
    if (a->inf) return 1;
    fq_t y2, x3, four;
    uint64_t f[6] = {4, 0, 0, 0, 0, 0};
    ora_fq_from_canonical(&four, f);
    ora_fq_mul(&y2, &a->y, &a->y);
    ora_fq_mul(&x3, &a->x, &a->x);
    ora_fq_mul(&x3, &x3, &a->x);
    ora_fq_add(&x3, &x3, &four);
    return fq_eq(&y2, &x3);
}
/* Montgomery batch inversion (what CurveGroup::normalize_batch does) */
void ora_g1_batch_to_affine(g1_affine_t *o, const g1_jac_t *a, size_t n) {
    fq_t *pref = (fq_t *)malloc((n + 1) * sizeof(fq_t));
    fq_t acc;
    fq_one(&acc);
    for (size_t i = 0; i < n; ++i) {
        pref[i] = acc;
        if (!fq_is_zero(&a[i].z)) ora_fq_mul(&acc, &acc, &a[i].z);
    }
    fq_t inv;
    ora_fq_inv(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fq_is_zero(&a[i].z)) {
            fq_zero(&o[i].x); fq_zero(&o[i].y); o[i].inf = 1;
            continue;
        }
        fq_t zi, zi2, zi3;
        ora_fq_mul(&zi, &inv, &pref[i]);
        ora_fq_mul(&inv, &inv, &a[i].z);
        ora_fq_mul(&zi2, &zi, &zi);
        ora_fq_mul(&zi3, &zi2, &zi);
        ora_fq_mul(&o[i].x, &a[i].x, &zi2);
        ora_fq_mul(&o[i].y, &a[i].y, &zi3);
        o[i].inf = 0;
    }
    free(pref);
}

/* kzg/src/utils.rs:19-40 over polynomial/src/utils.rs:141-157: for hypercube vertex i
 * (bits MSB first), prod_j (bit_j ? tau_j : 1 - tau_j). */
void ora_kzg_eq_points(fr_t *out, const fr_t *tau, size_t n_vars) {
    fr_t one;
    ora_fr_one(&one);
    for (size_t i = 0; i < ((size_t)1 << n_vars); ++i) {
        fr_t acc = one;
        for (size_t j = 0; j < n_vars; ++j) {
            int bit = (int)((i >> (n_vars - 1 - j)) & 1);
            if (!bit) {
                fr_t t;
                ora_fr_sub(&t, &one, &tau[j]);
                ora_fr_mul(&acc, &acc, &t);
            } else {
                ora_fr_mul(&acc, &acc, &tau[j]);
            }
        }
        out[i] = acc;
    }
}

/* trusted_setup.rs:25-35 */
void ora_kzg_multilinear_srs_g1(g1_jac_t *out, const fr_t *tau, size_t n_vars) {
    size_t n = (size_t)1 << n_vars;
    fr_t *pts = (fr_t *)malloc(n * sizeof(fr_t));
    ora_kzg_eq_points(pts, tau, n_vars);
    g1_jac_t g;
    ora_g1_generator(&g);
    for (size_t i = 0; i < n; ++i) {
        uint64_t c[4];
        ora_fr_to_canonical(c, &pts[i]);
        ora_g1_mul_bigint(&out[i], &g, c, 4);
    }
    free(pts);
}

/* univariate_kzg.rs:18-35 : G * tau^i for i in 0..=max_degree */
void ora_kzg_univariate_srs_g1(g1_jac_t *out, const fr_t *tau, size_t max_degree) {
    g1_jac_t g;
    ora_g1_generator(&g);
    for (size_t i = 0; i <= max_degree; ++i) {
        fr_t pw;
        uint64_t c[4];
        ora_fr_pow_u64(&pw, tau, (uint64_t)i);
        ora_fr_to_canonical(c, &pw);
        ora_g1_mul_bigint(&out[i], &g, c, 4);
    }
}

/* multilinear_kzg.rs:33-48 (require_equal_len = 1: the assert_eq! at :36-41) and
 * univariate_kzg.rs:37-58 (require_equal_len = 0; indexing srs[i] past the end panics -> -2). */
int ora_kzg_commitment(g1_jac_t *out, const fr_t *coeffs, size_t n_coeffs, const g1_jac_t *srs, size_t n_srs,
                       int require_equal_len) {
    if (require_equal_len && n_coeffs != n_srs) return -1;
    if (n_coeffs > n_srs) return -2;
    g1_jac_t acc;
    ora_g1_identity(&acc);
    for (size_t i = 0; i < n_coeffs; ++i) {
        uint64_t c[4];
        g1_jac_t t;
        ora_fr_to_canonical(c, &coeffs[i]);
        ora_g1_mul_bigint(&t, &srs[i], c, 4);
        ora_g1_add(&acc, &acc, &t);
    }
    *out = acc;
    return 0;
}

/* MultilinearKZG::open, multilinear_kzg.rs:50-88, step by step as the reference does it: every round commits the
 * quotient blown up to all n variables against the WHOLE srs (n naive commitments of 2^n terms each).
 *   get_poly_quotient  kzg/src/utils.rs:12-17   f(1, .) - f(0, .)
 *   get_poly_remainder kzg/src/utils.rs:5-10    f(z, .)
 *   add_to_front       evaluation_form.rs:86-96 (the table repeated 2 * 2^variable_length times)
 *   duplicate_evaluation :112-119, used in the last round with add_to_front(variable_index - 1)
 * Returns -1 on the shape panics (point count, srs length), -2 when n_vars < 2 (`variable_index - 1` underflows),
 * -3 on "Evaluation and final remainder mismatch!". */
int ora_kzg_open(fr_t *evaluation, g1_jac_t *proofs, const fr_t *evals, size_t n, const fr_t *points, size_t n_points,
                 const g1_jac_t *srs, size_t n_srs) {
    size_t n_vars = 0;
    while (((size_t)1 << n_vars) < n) ++n_vars;
    if (((size_t)1 << n_vars) != n || n_points != n_vars || n_srs != n) return -1;
    if (n_vars < 2) return -2;
    if (ora_mle_evaluation(evaluation, evals, n, points, n_points) != 0) return -1;
    fr_t *poly = (fr_t *)malloc(n * sizeof(fr_t));
    fr_t *f1 = (fr_t *)malloc(n * sizeof(fr_t)), *f0 = (fr_t *)malloc(n * sizeof(fr_t));
    fr_t *blown = (fr_t *)malloc(n * sizeof(fr_t));
    memcpy(poly, evals, n * sizeof(fr_t));
    fr_t one, zero, final_remainder;
    ora_fr_one(&one);
    ora_fr_zero(&zero);
    ora_fr_zero(&final_remainder);
    size_t cn = n;
    int rc = 0;
    for (size_t i = 0; i < n_vars; ++i) {
        ora_mle_partial_evaluation(f1, poly, cn, &one, 0);
        ora_mle_partial_evaluation(f0, poly, cn, &zero, 0);
        const size_t q = cn / 2;
        for (size_t j = 0; j < q; ++j) ora_fr_sub(&f1[j], &f1[j], &f0[j]);   /* quotient */
        size_t base_len, reps;
        if (i != n_vars - 1) {
            base_len = q;
            reps = (size_t)2 << i;                    /* add_to_front(&i): 2 * 2^i copies */
        } else {
            base_len = 2 * q;                         /* duplicate_evaluation, then add_to_front(&(i - 1)) */
            f1[1] = f1[0];
            reps = (size_t)2 << (i - 1);
            ora_mle_evaluation(&final_remainder, poly, cn, &points[i], 1);
        }
        for (size_t r = 0; r < reps; ++r) memcpy(&blown[r * base_len], f1, base_len * sizeof(fr_t));
        if (reps * base_len != n || ora_kzg_commitment(&proofs[i], blown, n, srs, n_srs, 1) != 0) { rc = -1; break; }
        if (i != n_vars - 1) {
            ora_mle_partial_evaluation(f0, poly, cn, &points[i], 0);      /* remainder */
            memcpy(poly, f0, q * sizeof(fr_t));
        }
        cn = q;
    }
    if (rc == 0 && !ora_fr_eq(evaluation, &final_remainder)) rc = -3;
    free(poly); free(f1); free(f0); free(blown);
    return rc;
}

/* UnivariateKZG::open (univariate_kzg.rs:60-81): evaluation = poly(z); numerator = poly - z (sic: the evaluation POINT is
 * subtracted, dense_univariate.rs:332-346 -- the quotient does not depend on that constant); quotient = numerator /
 * (x - z); proof = sum_i srs[i] * quotient[i].  Returns -2 when the quotient is longer than the srs (index panic). */
int ora_univariate_kzg_open(fr_t *evaluation, g1_jac_t *proof, const fr_t *coeffs, size_t n, const fr_t *z,
                            const g1_jac_t *srs, size_t n_srs) {
    ora_dense_evaluate(evaluation, coeffs, n, z);
    fr_t den[2];
    ora_fr_neg(&den[0], z);
    ora_fr_one(&den[1]);
    fr_t *num = (fr_t *)malloc((n + 1) * sizeof(fr_t));
    size_t nn = n;
    if (n == 0) { num[0] = *z; nn = 1; }                            /* Sub<F> on the zero polynomial returns [other] (:337-339) */
    else { memcpy(num, coeffs, n * sizeof(fr_t)); ora_fr_sub(&num[0], &num[0], z); }
    fr_t *q = (fr_t *)malloc((nn + 1) * sizeof(fr_t)), *r = (fr_t *)malloc((nn + 1) * sizeof(fr_t));
    size_t nq = 0, nr = 0;
    int rc = ora_dense_divide(q, &nq, r, &nr, num, nn, den, 2);
    if (rc == 0) rc = ora_kzg_commitment(proof, q, nq, srs, n_srs, 0);
    free(num); free(q); free(r);
    return rc;
}

/* CPU bucket-method MSM: NOT the reference's algorithm (the reference is the naive
 * sum above); provided so that large-size GPU results can be cross-checked in
 * seconds and as a context number.  Unsigned windows of c bits. */
void ora_msm_pippenger(g1_jac_t *out, const fr_t *scalars, const g1_affine_t *pts, size_t n) {
    unsigned c = 4;
    while (((size_t)1 << (c + 3)) < n && c < 16) ++c;
    size_t n_buckets = ((size_t)1 << c) - 1;
    unsigned n_windows = (255 + c - 1) / c;
    uint64_t *canon = (uint64_t *)malloc(n * 4 * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) ora_fr_to_canonical(&canon[4 * i], &scalars[i]);
    g1_jac_t *buckets = (g1_jac_t *)malloc(n_buckets * sizeof(g1_jac_t));
    g1_jac_t total;
    ora_g1_identity(&total);
    for (int w = (int)n_windows - 1; w >= 0; --w) {
        for (unsigned k = 0; k < c; ++k) ora_g1_double(&total, &total);
        for (size_t b = 0; b < n_buckets; ++b) ora_g1_identity(&buckets[b]);
        unsigned lo = (unsigned)w * c;
        for (size_t i = 0; i < n; ++i) {
            if (pts[i].inf) continue;
            uint64_t d = 0;
            for (unsigned k = 0; k < c; ++k) {
                unsigned bit = lo + k;
                if (bit < 256) d |= ((canon[4 * i + bit / 64] >> (bit % 64)) & 1) << k;
            }
            if (!d) continue;
            g1_jac_t pj;
            ora_g1_from_affine(&pj, &pts[i]);
            ora_g1_add(&buckets[d - 1], &buckets[d - 1], &pj);
        }
        g1_jac_t running, sum;
        ora_g1_identity(&running);
        ora_g1_identity(&sum);
        for (size_t b = n_buckets; b-- > 0;) {
            ora_g1_add(&running, &running, &buckets[b]);
            ora_g1_add(&sum, &sum, &running);
        }
        ora_g1_add(&total, &total, &sum);
    }
    *out = total;
    free(canon); free(buckets);
}

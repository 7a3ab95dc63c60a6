/*
 * field.c -- CPU ORACLE (test infrastructure): BLS12-381 Fr / Fq Montgomery
 * arithmetic, restating ark-ff 0.4.2 `Fp<MontBackend<_, N>, N>` as reached from
 * polynomial/src/multilinear/evaluation_form.rs:133 (mul/add/sub),
 * :58 (into_bigint().to_bytes_be()), transcripts/fiat-shamir/src/fiat_shamir.rs:28
 * (from_be_bytes_mod_order) and polynomial/src/univariate/domain.rs:38-40
 * (get_root_of_unity, inverse).  ark-ff is a crates.io dependency (^0.4.2), not
 * vendored in the reference; the algorithm restated is textbook CIOS Montgomery
 * with fully reduced outputs, so limb values are canonical and comparable.
 */
#include "zkoracle.h"
#include <string.h>

typedef unsigned __int128 u128;

typedef struct {
    int n;
    uint64_t p[6];    /* modulus */
    uint64_t r1[6];   /* R mod p  (Montgomery one) */
    uint64_t r2[6];   /* R^2 mod p */
    uint64_t inv;     /* -p^{-1} mod 2^64 */
} mont_t;

static const mont_t FR = {
    4,
    {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL, 0, 0},
    {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL, 0, 0},
    {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL, 0, 0},
    0xfffffffeffffffffULL};

static const mont_t FQ = {
    6,
    {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL,
     0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL},
    {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL, 0x77ce585370525745ULL,
     0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL},
    {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL, 0x67eb88a9939d83c0ULL,
     0x9a793e85b519952dULL, 0x11988fe592cae3aaULL},
    0x89f3fffcfffcfffdULL};

static inline int geq(const uint64_t *a, const uint64_t *b, int n) {
    for (int i = n - 1; i >= 0; --i) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline uint64_t sub_n(uint64_t *o, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t borrow = 0;
    for (int i = 0; i < n; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        o[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static inline uint64_t add_n(uint64_t *o, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t carry = 0;
    for (int i = 0; i < n; ++i) {
        u128 s = (u128)a[i] + b[i] + carry;
        o[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    return carry;
}

static void m_add(uint64_t *o, const uint64_t *a, const uint64_t *b, const mont_t *M) {
    uint64_t t[6];
    uint64_t c = add_n(t, a, b, M->n);
    if (c || geq(t, M->p, M->n)) sub_n(t, t, M->p, M->n);
    memcpy(o, t, 8 * M->n);
}
static void m_sub(uint64_t *o, const uint64_t *a, const uint64_t *b, const mont_t *M) {
    uint64_t t[6];
    if (sub_n(t, a, b, M->n)) add_n(t, t, M->p, M->n);
    memcpy(o, t, 8 * M->n);
}
/* CIOS Montgomery product: o = a*b*R^-1 mod p, fully reduced.  Fixed-width bodies so the
 * compiler unrolls them (this is also the CPU baseline; keep it an honest scalar port). */
#define DEFINE_MONT_MUL(NAME, NL)                                                              \
    static inline void NAME(uint64_t *o, const uint64_t *a, const uint64_t *b, const mont_t *M) { \
        uint64_t t[NL + 2] = {0};                                                              \
        for (int i = 0; i < NL; ++i) {                                                         \
            uint64_t c = 0;                                                                    \
            for (int j = 0; j < NL; ++j) {                                                     \
                u128 s = (u128)a[j] * b[i] + t[j] + c;                                         \
                t[j] = (uint64_t)s;                                                            \
                c = (uint64_t)(s >> 64);                                                       \
            }                                                                                  \
            u128 s = (u128)t[NL] + c;                                                          \
            t[NL] = (uint64_t)s;                                                               \
            t[NL + 1] = (uint64_t)(s >> 64);                                                   \
            uint64_t m = t[0] * M->inv;                                                        \
            s = (u128)m * M->p[0] + t[0];                                                      \
            c = (uint64_t)(s >> 64);                                                           \
            for (int j = 1; j < NL; ++j) {                                                     \
                s = (u128)m * M->p[j] + t[j] + c;                                              \
                t[j - 1] = (uint64_t)s;                                                        \
                c = (uint64_t)(s >> 64);                                                       \
            }                                                                                  \
            s = (u128)t[NL] + c;                                                               \
            t[NL - 1] = (uint64_t)s;                                                           \
            t[NL] = t[NL + 1] + (uint64_t)(s >> 64);                                           \
        }                                                                                      \
        if (t[NL] || geq(t, M->p, NL)) sub_n(t, t, M->p, NL);                                  \
        for (int i = 0; i < NL; ++i) o[i] = t[i];                                              \
    }
DEFINE_MONT_MUL(m_mul4, 4)
DEFINE_MONT_MUL(m_mul6, 6)
static inline void m_mul(uint64_t *o, const uint64_t *a, const uint64_t *b, const mont_t *M) {
    if (M->n == 4) m_mul4(o, a, b, M); else m_mul6(o, a, b, M);
}
static int m_is_zero(const uint64_t *a, int n) {
    uint64_t x = 0;
    for (int i = 0; i < n; ++i) x |= a[i];
    return x == 0;
}
/* o = a^e where e is a little-endian limb array (square-and-multiply, MSB first) */
static void m_pow(uint64_t *o, const uint64_t *a, const uint64_t *e, int e_limbs, const mont_t *M) {
    uint64_t acc[6], base[6];
    memcpy(acc, M->r1, 8 * M->n);
    memcpy(base, a, 8 * M->n);
    for (int i = e_limbs * 64 - 1; i >= 0; --i) {
        m_mul(acc, acc, acc, M);
        if ((e[i / 64] >> (i % 64)) & 1) m_mul(acc, acc, base, M);
    }
    memcpy(o, acc, 8 * M->n);
}
/* inverse by Fermat: a^(p-2) */
static int m_inv(uint64_t *o, const uint64_t *a, const mont_t *M) {
    if (m_is_zero(a, M->n)) return 0;
    uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0};
    sub_n(e, M->p, two, M->n);
    m_pow(o, a, e, M->n, M);
    return 1;
}
static void m_to_canonical(uint64_t *o, const uint64_t *a, const mont_t *M) {
    uint64_t one[6] = {1, 0, 0, 0, 0, 0};
    m_mul(o, a, one, M);
}
static void m_from_canonical(uint64_t *o, const uint64_t *a, const mont_t *M) { m_mul(o, a, M->r2, M); }

/* ---- Fr ---------------------------------------------------------------- */
void ora_fr_add(fr_t *o, const fr_t *a, const fr_t *b) { m_add(o->l, a->l, b->l, &FR); }
void ora_fr_sub(fr_t *o, const fr_t *a, const fr_t *b) { m_sub(o->l, a->l, b->l, &FR); }
void ora_fr_mul(fr_t *o, const fr_t *a, const fr_t *b) { m_mul4(o->l, a->l, b->l, &FR); }
void ora_fr_neg(fr_t *o, const fr_t *a) {
    fr_t z = {{0, 0, 0, 0}};
    m_sub(o->l, z.l, a->l, &FR);
}
int ora_fr_inv(fr_t *o, const fr_t *a) { return m_inv(o->l, a->l, &FR); }
void ora_fr_pow_u64(fr_t *o, const fr_t *a, uint64_t e) { m_pow(o->l, a->l, &e, 1, &FR); }
void ora_fr_from_u64(fr_t *o, uint64_t v) {
    uint64_t c[4] = {v, 0, 0, 0};
    m_from_canonical(o->l, c, &FR);
}
void ora_fr_one(fr_t *o) { memcpy(o->l, FR.r1, 32); }
void ora_fr_zero(fr_t *o) { memset(o->l, 0, 32); }
int ora_fr_is_zero(const fr_t *a) { return m_is_zero(a->l, 4); }
int ora_fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a->l, b->l, 32) == 0; }
void ora_fr_to_canonical(uint64_t out[4], const fr_t *a) { m_to_canonical(out, a->l, &FR); }
void ora_fr_from_canonical(fr_t *o, const uint64_t in[4]) { m_from_canonical(o->l, in, &FR); }

/* sumcheck/src/utils.rs:7-9 : element.into_bigint().to_bytes_be() -- 32 bytes, canonical, big-endian */
void ora_fr_to_bytes_be(uint8_t out[32], const fr_t *a) {
    uint64_t c[4];
    m_to_canonical(c, a->l, &FR);
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[31 - (8 * i + b)] = (uint8_t)(c[i] >> (8 * b));
}

/* fiat_shamir.rs:28 : F::from_be_bytes_mod_order(bytes) == int(bytes, big-endian) mod r.
 * Horner over bytes: acc = acc*256 + byte, all in the field. */
void ora_fr_from_be_bytes_mod_order(fr_t *o, const uint8_t *bytes, size_t len) {
    fr_t acc, k256, t;
    ora_fr_zero(&acc);
    ora_fr_from_u64(&k256, 256);
    for (size_t i = 0; i < len; ++i) {
        ora_fr_mul(&acc, &acc, &k256);
        ora_fr_from_u64(&t, bytes[i]);
        ora_fr_add(&acc, &acc, &t);
    }
    *o = acc;
}

/* domain.rs:38 : F::get_root_of_unity(n) for n a power of two:
 * omega = TWO_ADIC_ROOT_OF_UNITY (= 7^((r-1)/2^32)), squared (32 - log2 n) times. */
int ora_fr_get_root_of_unity(fr_t *o, uint64_t n) {
    if (n == 0 || (n & (n - 1))) return 0;
    int log_n = 0;
    while ((1ULL << log_n) < n) ++log_n;
    if (log_n > 32) return 0;
    fr_t g;
    ora_fr_from_u64(&g, 7);
    /* (r-1) >> 32 */
    uint64_t e[4];
    uint64_t rm1[4] = {FR.p[0] - 1, FR.p[1], FR.p[2], FR.p[3]};
    for (int i = 0; i < 4; ++i) e[i] = (rm1[i] >> 32) | (i < 3 ? (rm1[i + 1] << 32) : 0);
    fr_t w;
    m_pow(w.l, g.l, e, 4, &FR);
    for (int i = log_n; i < 32; ++i) ora_fr_mul(&w, &w, &w);
    *o = w;
    return 1;
}

/* ---- Fq ---------------------------------------------------------------- */
void ora_fq_add(fq_t *o, const fq_t *a, const fq_t *b) { m_add(o->l, a->l, b->l, &FQ); }
void ora_fq_sub(fq_t *o, const fq_t *a, const fq_t *b) { m_sub(o->l, a->l, b->l, &FQ); }
void ora_fq_mul(fq_t *o, const fq_t *a, const fq_t *b) { m_mul6(o->l, a->l, b->l, &FQ); }
int ora_fq_inv(fq_t *o, const fq_t *a) { return m_inv(o->l, a->l, &FQ); }
void ora_fq_to_canonical(uint64_t out[6], const fq_t *a) { m_to_canonical(out, a->l, &FQ); }
void ora_fq_from_canonical(fq_t *o, const uint64_t in[6]) { m_from_canonical(o->l, in, &FQ); }

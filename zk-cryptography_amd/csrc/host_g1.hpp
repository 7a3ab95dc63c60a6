// host_g1.hpp -- host-side epilogue of the MSM (product code, not the test oracle).
//
// The bucket method leaves (#windows x #terms) G1 points, each to be multiplied by a power of two and
// summed.  That is one 255-step double-and-add chain -- strictly serial, a few hundred group operations
// -- so it runs on the host CPU while the GPU is idle (as the GPU MSM libraries in production do), and
// it also converts the result to affine coordinates, the form parity with the reference is defined on.
// Plain 6 x 64-bit Montgomery arithmetic over BLS12-381 Fq with unsigned __int128.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace zkhost {

typedef unsigned __int128 u128;
struct Fq { uint64_t l[6]; };

static const uint64_t FQ_P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t FQ_R1[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                                  0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
static const uint64_t FQ_INV = 0x89f3fffcfffcfffdULL;

inline bool fq_is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
inline bool fq_eq(const Fq& a, const Fq& b) { return memcmp(a.l, b.l, 48) == 0; }
inline Fq fq_zero() { Fq z; memset(z.l, 0, 48); return z; }
inline Fq fq_one() { Fq o; memcpy(o.l, FQ_R1, 48); return o; }
inline bool geq_p(const uint64_t* t) {
    for (int i = 5; i >= 0; --i) { if (t[i] > FQ_P[i]) return true; if (t[i] < FQ_P[i]) return false; }
    return true;
}
inline void sub_p(uint64_t* t) {
    uint64_t br = 0;
    for (int i = 0; i < 6; ++i) { u128 d = (u128)t[i] - FQ_P[i] - br; t[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline Fq fq_add(const Fq& a, const Fq& b) {
    Fq r; uint64_t c = 0;
    for (int i = 0; i < 6; ++i) { u128 s = (u128)a.l[i] + b.l[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    if (geq_p(r.l)) sub_p(r.l);
    return r;
}
inline Fq fq_sub(const Fq& a, const Fq& b) {
    Fq r; uint64_t br = 0;
    for (int i = 0; i < 6; ++i) { u128 d = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) { uint64_t c = 0; for (int i = 0; i < 6; ++i) { u128 s = (u128)r.l[i] + FQ_P[i] + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
    return r;
}
inline Fq fq_dbl(const Fq& a) { return fq_add(a, a); }
inline Fq fq_mul(const Fq& a, const Fq& b) {
    uint64_t t[8] = {0};
    for (int i = 0; i < 6; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < 6; ++j) { u128 s = (u128)a.l[j] * b.l[i] + t[j] + c; t[j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        u128 s = (u128)t[6] + c; t[6] = (uint64_t)s; t[7] = (uint64_t)(s >> 64);
        uint64_t m = t[0] * FQ_INV;
        s = (u128)m * FQ_P[0] + t[0]; c = (uint64_t)(s >> 64);
        for (int j = 1; j < 6; ++j) { s = (u128)m * FQ_P[j] + t[j] + c; t[j - 1] = (uint64_t)s; c = (uint64_t)(s >> 64); }
        s = (u128)t[6] + c; t[5] = (uint64_t)s; t[6] = t[7] + (uint64_t)(s >> 64);
    }
    if (t[6] || geq_p(t)) sub_p(t);
    Fq r; memcpy(r.l, t, 48); return r;
}
inline Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }
// a^-1 (Montgomery form in, Montgomery form out; 0 -> 0).  Binary extended Euclid on the stored integer a R -- ~2 x 381 steps of shifts
// and subtractions on six limbs, ~10 us, against 384 squarings + ~190 products (~40 us) for a^(p-2): it ends every commitment's and
// every opening round's host epilogue -- then one product by R^3 takes (a R)^-1 = a^-1 R^-1 back to a^-1 R.
inline Fq fq_inv(const Fq& a) {
    if (fq_is_zero(a)) return a;
    static const uint64_t R2[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                                   0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
    static const Fq R3 = [] { Fq r2; memcpy(r2.l, R2, 48); return fq_mul(r2, r2); }();
    uint64_t u[6], v[6], x1[6] = {1, 0, 0, 0, 0, 0}, x2[6] = {0, 0, 0, 0, 0, 0};
    memcpy(u, a.l, 48);
    memcpy(v, FQ_P, 48);
    auto is_one = [](const uint64_t* t) { return t[0] == 1 && (t[1] | t[2] | t[3] | t[4] | t[5]) == 0; };
    auto shr1 = [](uint64_t* t) { for (int i = 0; i < 5; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 63); t[5] >>= 1; };
    auto halve = [&](uint64_t* t) {           // t / 2 mod p (t < p: t + p < 2^382)
        if (t[0] & 1) { uint64_t c = 0; for (int i = 0; i < 6; ++i) { u128 s = (u128)t[i] + FQ_P[i] + c; t[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
        shr1(t);
    };
    auto geq = [](const uint64_t* x, const uint64_t* y) { for (int i = 5; i >= 0; --i) { if (x[i] > y[i]) return true; if (x[i] < y[i]) return false; } return true; };
    auto sub = [](uint64_t* x, const uint64_t* y) { uint64_t br = 0; for (int i = 0; i < 6; ++i) { u128 d = (u128)x[i] - y[i] - br; x[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return br; };
    auto submod = [&](uint64_t* x, const uint64_t* y) {
        if (sub(x, y)) { uint64_t c = 0; for (int i = 0; i < 6; ++i) { u128 s = (u128)x[i] + FQ_P[i] + c; x[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
    };
    while (!is_one(u) && !is_one(v)) {
        while (!(u[0] & 1)) { shr1(u); halve(x1); }
        while (!(v[0] & 1)) { shr1(v); halve(x2); }
        if (geq(u, v)) { sub(u, v); submod(x1, x2); } else { sub(v, u); submod(x2, x1); }
    }
    Fq t;
    memcpy(t.l, is_one(u) ? x1 : x2, 48);
    return fq_mul(t, R3);
}

struct Xyzz { Fq x, y, zz, zzz; };
inline Xyzz xyzz_identity() { Xyzz p; p.x = p.y = p.zz = p.zzz = fq_zero(); return p; }
inline bool is_identity(const Xyzz& p) { return fq_is_zero(p.zz); }
inline Xyzz xyzz_double(const Xyzz& p) {   // dbl-2008-s-1
    if (is_identity(p)) return p;
    Xyzz r;
    Fq u = fq_dbl(p.y), v = fq_sqr(u), w = fq_mul(u, v), s = fq_mul(p.x, v), xx = fq_sqr(p.x);
    Fq m = fq_add(fq_dbl(xx), xx);
    r.x = fq_sub(fq_sqr(m), fq_dbl(s));
    r.y = fq_sub(fq_mul(m, fq_sub(s, r.x)), fq_mul(w, p.y));
    r.zz = fq_mul(v, p.zz);
    r.zzz = fq_mul(w, p.zzz);
    return r;
}
inline Xyzz xyzz_add(const Xyzz& a, const Xyzz& b) {   // add-2008-s, complete
    if (is_identity(b)) return a;
    if (is_identity(a)) return b;
    Fq u1 = fq_mul(a.x, b.zz), u2 = fq_mul(b.x, a.zz), s1 = fq_mul(a.y, b.zzz), s2 = fq_mul(b.y, a.zzz);
    Fq p = fq_sub(u2, u1), r = fq_sub(s2, s1);
    if (fq_is_zero(p)) return fq_is_zero(r) ? xyzz_double(a) : xyzz_identity();
    Fq pp = fq_sqr(p), ppp = fq_mul(p, pp), q = fq_mul(u1, pp);
    Xyzz o;
    o.x = fq_sub(fq_sub(fq_sqr(r), ppp), fq_dbl(q));
    o.y = fq_sub(fq_mul(r, fq_sub(q, o.x)), fq_mul(s1, ppp));
    o.zz = fq_mul(fq_mul(a.zz, b.zz), pp);
    o.zzz = fq_mul(fq_mul(a.zzz, b.zzz), ppp);
    return o;
}
inline Xyzz xyzz_from_affine(const uint64_t* xy, bool inf) {
    if (inf) return xyzz_identity();
    Xyzz p; memcpy(p.x.l, xy, 48); memcpy(p.y.l, xy + 6, 48); p.zz = fq_one(); p.zzz = fq_one();
    return p;
}
// x = X/ZZ, y = Y/ZZZ ; returns false (and zeros) for the identity
inline bool xyzz_to_affine(const Xyzz& p, uint64_t* out_xy) {
    if (is_identity(p)) { memset(out_xy, 0, 96); return false; }
    Fq inv = fq_inv(fq_mul(p.zz, p.zzz));            // one inversion: 1/(ZZ*ZZZ)
    Fq x = fq_mul(p.x, fq_mul(inv, p.zzz));          // X / ZZ
    Fq y = fq_mul(p.y, fq_mul(inv, p.zz));           // Y / ZZZ
    memcpy(out_xy, x.l, 48); memcpy(out_xy + 6, y.l, 48);
    return true;
}

// sum_i 2^{exp_i} * P_i  by one descending double-and-add sweep
inline Xyzz weighted_sum_pow2(const std::vector<Xyzz>& pts, const std::vector<uint32_t>& exps) {
    uint32_t top = 0;
    for (uint32_t e : exps) if (e > top) top = e;
    std::vector<Xyzz> level(top + 1, xyzz_identity());
    for (size_t i = 0; i < pts.size(); ++i) level[exps[i]] = xyzz_add(level[exps[i]], pts[i]);
    Xyzz acc = xyzz_identity();
    for (int e = (int)top; e >= 0; --e) {
        acc = xyzz_double(acc);
        acc = xyzz_add(acc, level[e]);
    }
    return acc;
}

}  // namespace zkhost

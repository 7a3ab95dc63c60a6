// g1.hpp -- BLS12-381 G1 (y^2 = x^3 + 4) group law for gfx950, device side.
//
// Replaces the ark-ec `short_weierstrass::Projective<g1::Config>` additions the reference performs
// inside `mul_bigint` / `+=` / `sum` at kzg/src/univariate_kzg.rs:53-54 and
// kzg/src/multilinear_kzg.rs:46-47.  Accumulators are kept in XYZZ coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; ZZ = 0 is the identity): a mixed addition of an affine SRS point
// costs 8M + 2S, and the doubling / identity / inverse cases the bucket method runs into are handled
// explicitly.  Coordinates are algorithm-dependent; parity with the reference is on the affine result.
#pragma once
#include "fp.hpp"

namespace zk {

// The 12-limb Montgomery product is ~900 instructions; the group law below needs 10-14 of them.  Keeping the
// product out of line keeps every MSM kernel inside the instruction cache (arguments travel in VGPRs).
#ifndef ZK_FQ_INLINE
__device__ __noinline__ Fq fq_mul(Fq a, Fq b) { return a * b; }
#else
__device__ __forceinline__ Fq fq_mul(const Fq& a, const Fq& b) { return a * b; }
#endif
__device__ __forceinline__ Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }

struct G1Affine {   // 96 bytes in memory: x[6] y[6] (uint64 Montgomery limbs); infinity kept in a side array
    Fq x, y;
};
struct G1Xyzz {
    Fq x, y, zz, zzz;
    __device__ __forceinline__ static G1Xyzz identity() {
        G1Xyzz p;
        p.x = Fq::zero(); p.y = Fq::zero(); p.zz = Fq::zero(); p.zzz = Fq::zero();
        return p;
    }
    __device__ __forceinline__ bool is_identity() const { return zz.is_zero(); }
};

__device__ __forceinline__ Fq load_fq(const uint64_t* __restrict__ p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1], c = q[2];
    Fq r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x; r.l[9] = c.y; r.l[10] = c.z; r.l[11] = c.w;
    return r;
}
__device__ __forceinline__ void store_fq(uint64_t* __restrict__ p, const Fq& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    q[2] = make_uint4(v.l[8], v.l[9], v.l[10], v.l[11]);
}
__device__ __forceinline__ G1Affine load_affine(const uint64_t* __restrict__ pts, size_t i) {
    G1Affine a;
    a.x = load_fq(pts + 12 * i);
    a.y = load_fq(pts + 12 * i + 6);
    return a;
}
__device__ __forceinline__ G1Xyzz load_xyzz(const uint64_t* __restrict__ p, size_t i) {
    G1Xyzz r;
    r.x = load_fq(p + 24 * i); r.y = load_fq(p + 24 * i + 6);
    r.zz = load_fq(p + 24 * i + 12); r.zzz = load_fq(p + 24 * i + 18);
    return r;
}
__device__ __forceinline__ void store_xyzz(uint64_t* __restrict__ p, size_t i, const G1Xyzz& v) {
    store_fq(p + 24 * i, v.x); store_fq(p + 24 * i + 6, v.y);
    store_fq(p + 24 * i + 12, v.zz); store_fq(p + 24 * i + 18, v.zzz);
}

// 2 * (affine point)  (mdbl-2008-s-1, a = 0); the point is not the identity and y != 0 on this curve
__device__ __forceinline__ G1Xyzz g1_double_affine(const G1Affine& p) {
    G1Xyzz r;
    Fq u = p.y.dbl();
    Fq v = fq_sqr(u);
    Fq w = fq_mul(u, v);
    Fq s = fq_mul(p.x, v);
    Fq xx = fq_sqr(p.x);
    Fq m = xx.dbl() + xx;
    r.x = fq_sqr(m) - s.dbl();
    r.y = fq_mul(m, s - r.x) - fq_mul(w, p.y);
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2 * acc (dbl-2008-s-1, a = 0)
__device__ __forceinline__ G1Xyzz g1_double(const G1Xyzz& p) {
    if (p.is_identity()) return p;
    G1Xyzz r;
    Fq u = p.y.dbl();
    Fq v = fq_sqr(u);
    Fq w = fq_mul(u, v);
    Fq s = fq_mul(p.x, v);
    Fq xx = fq_sqr(p.x);
    Fq m = xx.dbl() + xx;
    r.x = fq_sqr(m) - s.dbl();
    r.y = fq_mul(m, s - r.x) - fq_mul(w, p.y);
    r.zz = fq_mul(v, p.zz);
    r.zzz = fq_mul(w, p.zzz);
    return r;
}

// acc += affine point (madd-2008-s), complete: identity accumulator, doubling and inverse handled.
// `neg` adds -p (signed bucket digits).
__device__ __forceinline__ void g1_madd(G1Xyzz& acc, const G1Affine& p_in, bool neg) {
    G1Affine p = p_in;
    if (neg) p.y = p.y.neg();
    if (acc.is_identity()) {
        acc.x = p.x; acc.y = p.y; acc.zz = Fq::one(); acc.zzz = Fq::one();
        return;
    }
    Fq u2 = fq_mul(p.x, acc.zz);
    Fq s2 = fq_mul(p.y, acc.zzz);
    Fq pp_ = u2 - acc.x;   // P
    Fq r = s2 - acc.y;     // R
    if (pp_.is_zero()) {
        if (r.is_zero()) acc = g1_double_affine(p);   // same point
        else acc = G1Xyzz::identity();                // inverse points
        return;
    }
    Fq pp = fq_sqr(pp_);
    Fq ppp = fq_mul(pp_, pp);
    Fq q = fq_mul(acc.x, pp);
    Fq x3 = fq_sqr(r) - ppp - q.dbl();
    Fq y3 = fq_mul(r, q - x3) - fq_mul(acc.y, ppp);
    acc.x = x3;
    acc.y = y3;
    acc.zz = fq_mul(acc.zz, pp);
    acc.zzz = fq_mul(acc.zzz, ppp);
}

// acc += q (add-2008-s), complete.
__device__ __forceinline__ void g1_add(G1Xyzz& acc, const G1Xyzz& q) {
    if (q.is_identity()) return;
    if (acc.is_identity()) { acc = q; return; }
    Fq u1 = fq_mul(acc.x, q.zz);
    Fq u2 = fq_mul(q.x, acc.zz);
    Fq s1 = fq_mul(acc.y, q.zzz);
    Fq s2 = fq_mul(q.y, acc.zzz);
    Fq p = u2 - u1;
    Fq r = s2 - s1;
    if (p.is_zero()) {
        if (r.is_zero()) acc = g1_double(acc);
        else acc = G1Xyzz::identity();
        return;
    }
    Fq pp = fq_sqr(p);
    Fq ppp = fq_mul(p, pp);
    Fq qq = fq_mul(u1, pp);
    Fq x3 = fq_sqr(r) - ppp - qq.dbl();
    Fq y3 = fq_mul(r, qq - x3) - fq_mul(s1, ppp);
    Fq zz3 = fq_mul(fq_mul(acc.zz, q.zz), pp);
    Fq zzz3 = fq_mul(fq_mul(acc.zzz, q.zzz), ppp);
    acc.x = x3; acc.y = y3; acc.zz = zz3; acc.zzz = zzz3;
}

}  // namespace zk

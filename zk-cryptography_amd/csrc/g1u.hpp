// g1u.hpp -- BLS12-381 G1 group law over the unsaturated field (fqu.hpp): the MSM's bucket arithmetic.
//
// Same XYZZ formulas as g1.hpp (madd-2008-s, add-2008-s, dbl-2008-s-1); every subtraction adds a fixed multiple of p
// and the comments track the value bounds (in multiples of p) that make those constants sufficient:
//     stored X < 14p, Y < 6p, ZZ, ZZZ < 2p; products < 2p; the identity is stored as all-zero limbs.
#pragma once
#include "fqu.hpp"

namespace zk {

struct G1AffineU { FqU x, y; };                  // x, y < 2p
struct G1XyzzU {
    FqU x, y, zz, zzz;
    __device__ __forceinline__ static G1XyzzU identity() {
        G1XyzzU p;
        p.x = FqU::zero(); p.y = FqU::zero(); p.zz = FqU::zero(); p.zzz = FqU::zero();
        return p;
    }
    __device__ __forceinline__ bool is_identity() const { return zz.all_zero(); }
};

// memory: affine = 2 x 64 B, XYZZ = 4 x 64 B
__device__ __forceinline__ G1AffineU load_affine_u(const uint32_t* __restrict__ p, size_t i) {
    G1AffineU a;
    a.x = load_fqu(p + 32 * i);
    a.y = load_fqu(p + 32 * i + 16);
    return a;
}
__device__ __forceinline__ G1XyzzU load_xyzz_u(const uint32_t* __restrict__ p, size_t i) {
    G1XyzzU r;
    r.x = load_fqu(p + 64 * i); r.y = load_fqu(p + 64 * i + 16);
    r.zz = load_fqu(p + 64 * i + 32); r.zzz = load_fqu(p + 64 * i + 48);
    return r;
}
__device__ __forceinline__ void store_xyzz_u(uint32_t* __restrict__ p, size_t i, const G1XyzzU& v) {
    store_fqu(p + 64 * i, v.x); store_fqu(p + 64 * i + 16, v.y);
    store_fqu(p + 64 * i + 32, v.zz); store_fqu(p + 64 * i + 48, v.zzz);
}

// 2 * (x, y), affine input (x, y < 4p).  mdbl-2008-s-1
__device__ __forceinline__ G1XyzzU g1u_double_affine(const G1AffineU& p) {
    G1XyzzU r;
    FqU u = fqu_dbl(p.y);                                   // < 8p, limbs < 2^29.1
    FqU v = fqu_sqr(u);                                     // < 2p
    FqU w = fqu_mul(u, v);
    FqU s = fqu_mul(p.x, v);
    FqU xx = fqu_sqr(p.x);
    FqU m = fqu_weak_norm(fqu_add(fqu_dbl(xx), xx));        // 3 xx < 6p
    r.x = fqu_sub<8>(fqu_sqr(m), fqu_dbl(s));               // < 2p + 8p   (2s < 4p)
    r.y = fqu_sub<4>(fqu_mul(m, fqu_sub<16>(s, r.x)), fqu_mul(w, p.y));   // < 6p
    r.zz = v;
    r.zzz = w;
    return r;
}
// 2 * acc.  dbl-2008-s-1
__device__ __forceinline__ G1XyzzU g1u_double(const G1XyzzU& p) {
    if (p.is_identity()) return p;
    G1XyzzU r;
    FqU u = fqu_dbl(p.y);                                   // < 12p
    FqU v = fqu_sqr(u);
    FqU w = fqu_mul(u, v);
    FqU s = fqu_mul(p.x, v);
    FqU xx = fqu_sqr(p.x);
    FqU m = fqu_weak_norm(fqu_add(fqu_dbl(xx), xx));
    r.x = fqu_sub<8>(fqu_sqr(m), fqu_dbl(s));               // < 10p
    r.y = fqu_sub<4>(fqu_mul(m, fqu_sub<16>(s, r.x)), fqu_mul(w, p.y));   // < 6p
    r.zz = fqu_mul(v, p.zz);
    r.zzz = fqu_mul(w, p.zzz);
    return r;
}

// acc += (+-) affine point.  madd-2008-s, complete.
__device__ __forceinline__ void g1u_madd(G1XyzzU& acc, const G1AffineU& p_in, bool neg) {
    G1AffineU p = p_in;
    if (neg) p.y = fqu_neg4(p.y);                           // 4p - y < 4p
    if (acc.is_identity()) {
        acc.x = p.x; acc.y = p.y; acc.zz = FqU::one(); acc.zzz = FqU::one();
        return;
    }
    FqU u2 = fqu_mul(p.x, acc.zz);
    FqU s2 = fqu_mul(p.y, acc.zzz);
    FqU pp_ = fqu_sub<16>(u2, acc.x);                       // P = U2 - X1 (+16p), X1 < 14p      -> < 18p
    FqU r = fqu_sub<8>(s2, acc.y);                          // R = S2 - Y1 (+8p),  Y1 < 6p       -> < 10p
    if (fqu_is_zero_mod_p(pp_)) {
        if (fqu_is_zero_mod_p(r)) acc = g1u_double_affine(p);   // same point
        else acc = G1XyzzU::identity();                         // inverse points
        return;
    }
    FqU pp = fqu_sqr(pp_);
    FqU ppp = fqu_mul(pp_, pp);
    FqU q = fqu_mul(acc.x, pp);
    FqU x3 = fqu_sub<8>(fqu_sub<4>(fqu_sqr(r), ppp), fqu_dbl(q));         // R^2 - PPP - 2Q  < 2p + 4p + 8p = 14p
    FqU y3 = fqu_sub<4>(fqu_mul(r, fqu_sub<16>(q, x3)), fqu_mul(acc.y, ppp));   // < 6p
    acc.x = x3;
    acc.y = y3;
    acc.zz = fqu_mul(acc.zz, pp);
    acc.zzz = fqu_mul(acc.zzz, ppp);
}

// acc += q.  add-2008-s, complete.
__device__ __forceinline__ void g1u_add(G1XyzzU& acc, const G1XyzzU& q) {
    if (q.is_identity()) return;
    if (acc.is_identity()) { acc = q; return; }
    FqU u1 = fqu_mul(acc.x, q.zz);
    FqU u2 = fqu_mul(q.x, acc.zz);
    FqU s1 = fqu_mul(acc.y, q.zzz);
    FqU s2 = fqu_mul(q.y, acc.zzz);
    FqU p = fqu_sub<4>(u2, u1);                             // < 6p
    FqU r = fqu_sub<4>(s2, s1);
    if (fqu_is_zero_mod_p(p)) {
        if (fqu_is_zero_mod_p(r)) acc = g1u_double(acc);
        else acc = G1XyzzU::identity();
        return;
    }
    FqU pp = fqu_sqr(p);
    FqU ppp = fqu_mul(p, pp);
    FqU qq = fqu_mul(u1, pp);
    FqU x3 = fqu_sub<8>(fqu_sub<4>(fqu_sqr(r), ppp), fqu_dbl(qq));        // < 14p
    FqU y3 = fqu_sub<4>(fqu_mul(r, fqu_sub<16>(qq, x3)), fqu_mul(s1, ppp));   // < 6p
    FqU zz3 = fqu_mul(fqu_mul(acc.zz, q.zz), pp);
    FqU zzz3 = fqu_mul(fqu_mul(acc.zzz, q.zzz), ppp);
    acc.x = x3; acc.y = y3; acc.zz = zz3; acc.zzz = zzz3;
}

// ---- one addition on FOUR lanes -----------------------------------------------------------------------------------
// The trees that end every reduction (msm_wave_tree_sum) add ever fewer pairs, and a wave pays an addition's ~7 k instructions for a
// level however few of its lanes still take part.  add-2008-s is four products deep with up to four independent products per
// level:   { X1 ZZ2, X2 ZZ1, Y1 ZZZ2, Y2 ZZZ1 }   { P^2, R^2, ZZ1 ZZ2, ZZZ1 ZZZ2 }   { P PP, U1 PP, (ZZ1 ZZ2) PP }   { R (Q - X3), S1 PPP, (ZZZ1 ZZZ2) PPP }
// so the four lanes of a quad, holding the same two points, each take one product of a level (operands picked by the lane's role,
// results handed round the quad by DPP): 4 products + ~1 k instructions of selects, moves and lazy additions instead of 14 products.
template <int K> __device__ __forceinline__ FqU fqu_quad_bcast(const FqU& v) {          // lane K of the quad's value, on all four
    FqU r;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], K * 0x55, 0xf, 0xf, false);
    return r;
}
__device__ __forceinline__ FqU fqu_sel4(uint32_t role, const FqU& a0, const FqU& a1, const FqU& a2, const FqU& a3) {
    FqU r;
#pragma unroll
    for (int i = 0; i < FqU::N; ++i) {
        const uint32_t lo = role & 1 ? a1.l[i] : a0.l[i], hi = role & 1 ? a3.l[i] : a2.l[i];
        r.l[i] = role & 2 ? hi : lo;
    }
    return r;
}
// a + b on every lane of a quad whose four lanes hold the same a, b (any quad of a wave; all 64 lanes call).  Complete like g1u_add;
// same formulas, same bounds on the stored coordinates.
__device__ __forceinline__ G1XyzzU g1u_add_quad(const G1XyzzU& a, const G1XyzzU& b) {
    const uint32_t role = threadIdx.x & 3;
    const bool a_id = a.is_identity(), b_id = b.is_identity();
    // level 1: u1 = X1 ZZ2 | u2 = X2 ZZ1 | s1 = Y1 ZZZ2 | s2 = Y2 ZZZ1
    const FqU m1 = fqu_mul(fqu_sel4(role, a.x, b.x, a.y, b.y), fqu_sel4(role, b.zz, a.zz, b.zzz, a.zzz));
    const FqU u1 = fqu_quad_bcast<0>(m1), u2 = fqu_quad_bcast<1>(m1), s1 = fqu_quad_bcast<2>(m1), s2 = fqu_quad_bcast<3>(m1);
    const FqU p = fqu_sub<4>(u2, u1);                       // < 6p
    const FqU r = fqu_sub<4>(s2, s1);
    if (!a_id && !b_id && fqu_is_zero_mod_p(p)) {          // the same point or inverse points (quad-uniform: the four lanes hold the same values)
        if (fqu_is_zero_mod_p(r)) return g1u_double(a);
        return G1XyzzU::identity();
    }
    // level 2: pp = P^2 | rr = R^2 | zz12 = ZZ1 ZZ2 | zzz12 = ZZZ1 ZZZ2
    const FqU m2 = fqu_mul(fqu_sel4(role, p, r, a.zz, a.zzz), fqu_sel4(role, p, r, b.zz, b.zzz));
    const FqU pp = fqu_quad_bcast<0>(m2), rr = fqu_quad_bcast<1>(m2);
    // level 3: ppp = P PP | qq = U1 PP | zz3 = zz12 PP | (idle: zzz12 carried along)
    const FqU m3 = fqu_mul(fqu_sel4(role, p, u1, m2, m2), pp);
    const FqU ppp = fqu_quad_bcast<0>(m3), qq = fqu_quad_bcast<1>(m3);
    const FqU x3 = fqu_sub<8>(fqu_sub<4>(rr, ppp), fqu_dbl(qq));          // < 14p
    // level 4: r (qq - x3) | s1 ppp | (idle) | zzz3 = zzz12 ppp
    const FqU m4 = fqu_mul(fqu_sel4(role, r, s1, r, m2), fqu_sel4(role, fqu_sub<16>(qq, x3), ppp, ppp, ppp));
    G1XyzzU o;
    o.x = x3;
    o.y = fqu_sub<4>(fqu_quad_bcast<0>(m4), fqu_quad_bcast<1>(m4));      // < 6p
    o.zz = fqu_quad_bcast<2>(m3);
    o.zzz = fqu_quad_bcast<3>(m4);
    if (b_id) return a;
    if (a_id) return b;
    return o;
}

}  // namespace zk

"""polynomial::ComposedMultilinear and the composed / multi-composed sumcheck provers on the GPU.

Mirrors polynomial/src/composed/composed_multilinear.rs:8-120,
sumcheck/src/composed/composed_sumcheck.rs:20-67 and
sumcheck/src/composed/multi_composed_sumcheck.rs:13-121 (provers only; verifiers are host-side, out of scope).
"""
import ctypes as C

import numpy as np

from zk_cryptography_amd import _native as N
from zk_cryptography_amd.field import Fr
from zk_cryptography_amd.polynomial import Multilinear

MAX_MONO = 7


class ComposedMultilinear:
    def __init__(self, polys):
        """ComposedMultilinear::new (composed_multilinear.rs:12-18): all tables share n_vars."""
        polys = [p if isinstance(p, Multilinear) else Multilinear(p) for p in polys]
        assert all(p.n_vars == polys[0].n_vars for p in polys)
        self.polys = polys

    def n_vars(self):
        return self.polys[0].n_vars

    def max_degree(self):
        """composed_multilinear.rs:101-103"""
        return len(self.polys)

    def partial_evaluation(self, evaluation_point, variable_index):
        """composed_multilinear.rs:63-75"""
        return ComposedMultilinear([p.partial_evaluation(evaluation_point, variable_index) for p in self.polys])

    def evaluation(self, points):
        """composed_multilinear.rs:51-61: product of the tables' evaluations -> python int (canonical)"""
        acc = 1
        for p in self.polys:
            acc = acc * Fr.to_ints(p.evaluation(points))[0] % Fr.MODULUS
        return acc

    def to_bytes(self):
        """composed_multilinear.rs:40-48"""
        return b"".join(p.to_bytes() for p in self.polys)

    def _element_wise(self, op):
        if not self.polys:
            raise IndexError("element_wise on an empty ComposedMultilinear")      # self.polys[0] panics
        first = self.polys[0]
        if any(len(p) < len(first) for p in self.polys):
            raise IndexError("a table is shorter than the first one")             # v.evaluations[i] panics
        out = first._new_like(len(first))
        N.check(N.lib().zkhip_composed_element_wise(first._ctx.handle, C.c_int(op), _ptr_array(self._ptrs()),
                                                    C.c_uint32(len(self.polys)), C.c_size_t(len(first)), N.ptr(out)),
                "composed_element_wise")
        return out

    def element_wise_product(self):
        """ComposedMultilinearTrait::element_wise_product (composed_multilinear.rs:105-111) -> device tensor int64 [n, 4]
        (the reference's Vec<F>)"""
        return self._element_wise(0)

    def element_wise_add(self):
        """ComposedMultilinearTrait::element_wise_add (composed_multilinear.rs:113-119)"""
        return self._element_wise(1)

    def _ptrs(self):
        return [p.evaluations.data_ptr() for p in self.polys]


def _ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*ptrs)


class ComposedSumcheckProof:
    """composed_sumcheck.rs:15-18: {poly, round_polys}; round_polys is uint64 [n_vars, K+1, 4]"""

    def __init__(self, poly, round_polys):
        self.poly = poly
        self.round_polys = round_polys


class ComposedSumcheck:
    def __init__(self, poly):
        """ComposedSumcheck::new (composed_sumcheck.rs:21-26)"""
        self.poly = poly
        self.sum = np.zeros(4, dtype=np.uint64)

    @staticmethod
    def calculate_poly_sum(poly):
        """composed_sumcheck.rs:28-30"""
        out = np.empty(4, dtype=np.uint64)
        ctx = N.Context.get(poly.polys[0].evaluations.device.index)
        N.check(N.lib().zkhip_composed_sum(ctx.handle, _ptr_array(poly._ptrs()), C.c_uint32(len(poly.polys)),
                                           C.c_size_t(len(poly.polys[0])), out.ctypes.data_as(C.c_void_p)), "composed_sum")
        return out

    def prove(self):
        """composed_sumcheck.rs:32-67 -> (ComposedSumcheckProof, challenges [n_vars, 4])"""
        k, nv = len(self.poly.polys), self.poly.n_vars()
        rp = np.empty((max(nv, 1), k + 1, 4), dtype=np.uint64)
        ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
        ctx = N.Context.get(self.poly.polys[0].evaluations.device.index)
        N.check(N.lib().zkhip_composed_prove(ctx.handle, _ptr_array(self.poly._ptrs()), C.c_uint32(k),
                                             C.c_size_t(len(self.poly.polys[0])), rp.ctypes.data_as(C.c_void_p),
                                             ch.ctypes.data_as(C.c_void_p)), "composed_prove")
        return ComposedSumcheckProof(self.poly, rp[:nv]), ch[:nv]


class SparseUnivariatePolynomial:
    """polynomial/src/univariate/sparse_univariate.rs:12-20: monomials (coeff, pow), both Montgomery uint64[4]."""

    def __init__(self, coeffs, pows):
        self.coeffs = coeffs
        self.pows = pows

    def monomials(self):
        return list(zip(Fr.to_ints(self.coeffs), Fr.to_ints(self.pows))) if len(self.coeffs) else []

    def to_bytes(self):
        """sparse_univariate.rs:27-34"""
        out = b""
        for c, p in self.monomials():
            out += c.to_bytes(32, "big") + p.to_bytes(32, "big")
        return out


class MultiComposedSumcheckProof:
    """multi_composed_sumcheck.rs:12-16 (named ComposedSumcheckProof there): {round_polys, sum}"""

    def __init__(self, round_polys, sum_):
        self._round_polys = round_polys
        self._packed = None
        self.sum = sum_

    @classmethod
    def from_packed(cls, monomials, lens, sum_):
        """The prover's output as it comes off the device -- monomials uint64 [n_rounds, MAX_MONO, 2, 4] (coeff, pow), lens
        [n_rounds] -- wrapped without building a SparseUnivariatePolynomial per round; `round_polys` builds them on first
        use (a depth-20 GKR proof has ~420 rounds, and host time between proofs is GPU idle time)."""
        self = cls(None, sum_)
        self._packed = (monomials, lens)
        return self

    @property
    def round_polys(self):
        if self._round_polys is None:
            mono, lens = self._packed
            self._round_polys = [SparseUnivariatePolynomial(mono[r, : lens[r], 0].copy(), mono[r, : lens[r], 1].copy()) for r in range(len(lens))]
        return self._round_polys

    @round_polys.setter
    def round_polys(self, v):
        self._round_polys = v

    def to_bytes(self):
        """multi_composed_sumcheck.rs:24-31"""
        return b"".join(rp.to_bytes() for rp in self.round_polys)


class MultiComposedSumcheckProver:
    @staticmethod
    def _flat(poly):
        ptrs, sizes = [], []
        for term in poly:
            ptrs += term._ptrs()
            sizes.append(len(term.polys))
        return ptrs, sizes

    @staticmethod
    def calculate_poly_sum(poly):
        """multi_composed_sumcheck.rs:36-45"""
        ptrs, sizes = MultiComposedSumcheckProver._flat(poly)
        out = np.empty(4, dtype=np.uint64)
        ctx = N.Context.get(poly[0].polys[0].evaluations.device.index)
        N.check(N.lib().zkhip_multi_composed_sum(ctx.handle, _ptr_array(ptrs), (C.c_uint32 * len(sizes))(*sizes),
                                                 C.c_uint32(len(sizes)), C.c_size_t(len(poly[0].polys[0])),
                                                 out.ctypes.data_as(C.c_void_p)), "multi_composed_sum")
        return out

    @staticmethod
    def _prove(poly, sum_, partial):
        ptrs, sizes = MultiComposedSumcheckProver._flat(poly)
        nv = poly[0].n_vars()
        lens = np.zeros(max(nv, 1), dtype=np.uint32)
        rp = np.zeros((max(nv, 1), MAX_MONO, 2, 4), dtype=np.uint64)
        ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
        s = np.ascontiguousarray(sum_, dtype=np.uint64).reshape(4)
        ctx = N.Context.get(poly[0].polys[0].evaluations.device.index)
        st = N.lib().zkhip_multi_composed_prove(ctx.handle, _ptr_array(ptrs), (C.c_uint32 * len(sizes))(*sizes),
                                                C.c_uint32(len(sizes)), C.c_size_t(len(poly[0].polys[0])),
                                                s.ctypes.data_as(C.c_void_p), C.c_int(1 if partial else 0),
                                                lens.ctypes.data_as(C.c_void_p), rp.ctypes.data_as(C.c_void_p),
                                                ch.ctypes.data_as(C.c_void_p))
        N.check(st, "multi_composed_prove")
        polys = [SparseUnivariatePolynomial(rp[r, : lens[r], 0].copy(), rp[r, : lens[r], 1].copy()) for r in range(nv)]
        return MultiComposedSumcheckProof(polys, s.copy()), ch[:nv]

    @staticmethod
    def prove(poly, sum_):
        """multi_composed_sumcheck.rs:47-54 -> Ok((proof, challenges))"""
        return MultiComposedSumcheckProver._prove(poly, sum_, False)

    @staticmethod
    def prove_partial(poly, sum_):
        """multi_composed_sumcheck.rs:56-62"""
        return MultiComposedSumcheckProver._prove(poly, sum_, True)

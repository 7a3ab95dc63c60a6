"""sumcheck::Sumcheck prover on the GPU (sumcheck/src/sumcheck.rs:17-61).

`prove` runs the whole round loop on the device -- fused half-sums + fold kernels, and a
device-resident SHA-256 Fiat-Shamir transcript -- and returns the same observable values as
the reference: the claimed sum, one 2-evaluation round polynomial per variable, and the
challenges.  Verifiers are host-side and out of scope.
"""
import ctypes as C

import numpy as np

from zk_cryptography_amd import _native as N
from zk_cryptography_amd.polynomial import Multilinear


class SumcheckProof:
    """sumcheck.rs:11-15: {poly, sum, univariate_poly}.  `poly` is a reference to the prover's table
    (the Rust struct clones it); univariate_poly is uint64 [n_vars, 2, 4]."""

    def __init__(self, poly, sum_, univariate_poly):
        self.poly = poly
        self.sum = sum_
        self.univariate_poly = univariate_poly


class Sumcheck:
    def __init__(self, poly):
        """Sumcheck::new (sumcheck.rs:18-23)"""
        self.poly = poly if isinstance(poly, Multilinear) else Multilinear(poly)
        self.sum = np.zeros(4, dtype=np.uint64)   # Default::default()
        self._block_sums = None                    # device tensor kept by poly_sum() for prove()
        self._log_blocks = 0

    def poly_sum(self):
        """sumcheck.rs:25-27.  One streaming pass; the block sums it produces on the way stay on the device so that
        prove() does not re-read the table for its first rounds."""
        import torch
        n = len(self.poly)
        if n == 1:
            self.sum = self.poly.to_numpy()[0].copy()
            return
        lb = N.lib().zkhip_sumcheck_plan_log_blocks(C.c_size_t(n))
        buf = torch.empty(((1 << lb) + 1, 4), dtype=torch.int64, device=self.poly.evaluations.device)
        tot = np.empty(4, dtype=np.uint64)
        N.check(N.lib().zkhip_mle_block_sums(self.poly._ctx.handle, N.ptr(self.poly.evaluations), C.c_size_t(n),
                                             C.c_uint32(lb), N.ptr(buf), tot.ctypes.data_as(C.c_void_p)), "block_sums")
        self._block_sums, self._log_blocks = (buf, lb) if lb else (None, 0)
        self.sum = tot

    def prove(self):
        """sumcheck.rs:29-61 -> (SumcheckProof, challenges uint64 [n_vars, 4]).

        The transcript absorbs `self.sum` exactly as the reference does (zero if poly_sum() was never
        called)."""
        nv = self.poly.n_vars
        s = np.empty(4, dtype=np.uint64)
        rp = np.empty((max(nv, 1), 2, 4), dtype=np.uint64)
        ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
        st = N.lib().zkhip_sumcheck_prove(self.poly._ctx.handle, N.ptr(self.poly.evaluations),
                                          C.c_size_t(len(self.poly)),
                                          np.ascontiguousarray(self.sum, dtype=np.uint64).ctypes.data_as(C.c_void_p),
                                          N.ptr(self._block_sums) if self._block_sums is not None else None,
                                          C.c_uint32(self._log_blocks), s.ctypes.data_as(C.c_void_p),
                                          rp.ctypes.data_as(C.c_void_p), ch.ctypes.data_as(C.c_void_p))
        N.check(st, "sumcheck_prove")
        return SumcheckProof(self.poly, s, rp[:nv]), ch[:nv]

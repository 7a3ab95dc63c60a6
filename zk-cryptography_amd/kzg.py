"""kzg::{TrustedSetup, MultilinearKZG, UnivariateKZG} -- the commit path on the GPU.

Mirrors kzg/src/interface.rs:10-55 for `commitment` (a Pippenger multi-scalar multiplication over
BLS12-381 G1), `open` (one commitment per variable against the folded SRS) and the G1 half of the trusted
setup.  The pairing `verify` is out of scope.  A commitment is returned as affine coordinates (x, y Montgomery limbs + infinity flag):
the reference's Jacobian representation is algorithm dependent, equality is defined on the affine point.
"""
import ctypes as C

import numpy as np

from zk_cryptography_amd import _native as N
from zk_cryptography_amd.polynomial import Multilinear, _fr_host, _to_device


class G1Affine:
    def __init__(self, xy, inf):
        self.xy = np.ascontiguousarray(xy, dtype=np.uint64).reshape(12)
        self.infinity = bool(inf)

    def __eq__(self, o):
        return isinstance(o, G1Affine) and self.infinity == o.infinity and (self.infinity or np.array_equal(self.xy, o.xy))

    def coords(self):
        """(x, y) as canonical python ints"""
        q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
        rinv = pow(pow(2, 384, q), -1, q)
        x = sum(int(self.xy[k]) << (64 * k) for k in range(6)) * rinv % q
        y = sum(int(self.xy[6 + k]) << (64 * k) for k in range(6)) * rinv % q
        return x, y


class DenseUnivariatePolynomial:
    """polynomial::DenseUnivariatePolynomial (dense_univariate.rs:15-17): coefficients, low degree first, in HBM."""

    def __init__(self, coefficients, device=None):
        self.coefficients = _to_device(coefficients, device)

    def __len__(self):
        return self.coefficients.shape[0]

    def is_zero(self):
        """dense_univariate.rs:41-43: an EMPTY coefficient vector"""
        return len(self) == 0

    def degree(self):
        """dense_univariate.rs:199-207: zero leading coefficients do not count; 0 for the zero polynomial"""
        d = C.c_size_t(0)
        if len(self):
            ctx = N.Context.get(self.coefficients.device.index)
            N.check(N.lib().zkhip_dense_degree(ctx.handle, N.ptr(self.coefficients), C.c_size_t(len(self)), C.byref(d)), "degree")
        return d.value

    def evaluate(self, point):
        """dense_univariate.rs:184-196 -> uint64[4] (Montgomery)"""
        out = np.zeros(4, dtype=np.uint64)
        if len(self):
            z = _fr_host(point).reshape(4)
            ctx = N.Context.get(self.coefficients.device.index)
            N.check(N.lib().zkhip_dense_evaluate(ctx.handle, N.ptr(self.coefficients), C.c_size_t(len(self)),
                                                 z.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)), "evaluate")
        return out

    def __mul__(self, other):
        """Mul (dense_univariate.rs:210-233): schoolbook product of the first degree()+1 coefficients in the reference,
        the same coefficients through three NTTs here; Mul<F> (:235-251) for a field element."""
        import torch
        if isinstance(other, DenseUnivariatePolynomial):
            if self.is_zero() or other.is_zero():
                return DenseUnivariatePolynomial(torch.empty((0, 4), dtype=torch.int64, device="cuda"))
            from zk_cryptography_amd.univariate import UnivariateEval
            a = DenseUnivariatePolynomial(self.coefficients[: self.degree() + 1])
            b = DenseUnivariatePolynomial(other.coefficients[: other.degree() + 1])
            return UnivariateEval.multiply(a, b)
        scalar = _fr_host(other).reshape(4)
        if self.is_zero() or not scalar.any():
            return DenseUnivariatePolynomial(torch.empty((0, 4), dtype=torch.int64, device="cuda"))
        return DenseUnivariatePolynomial((Multilinear._wrap(self.coefficients) * scalar).evaluations)


class TrustedSetup:
    """kzg/src/trusted_setup.rs:9-13, G1 side only, stored affine in HBM: points int64 [n, 12], inf uint8 [n]."""

    def __init__(self, points_xy, inf):
        import torch
        if not isinstance(points_xy, torch.Tensor):
            points_xy = torch.from_numpy(np.ascontiguousarray(points_xy, dtype=np.uint64).view(np.int64)).cuda()
            inf = torch.from_numpy(np.ascontiguousarray(inf, dtype=np.uint8)).cuda()
        self.powers_of_tau_in_g1 = points_xy.contiguous()
        self.inf = inf.contiguous()

    def __len__(self):
        return self.powers_of_tau_in_g1.shape[0]

    def _stamp(self):
        """Identity of the SRS the derived caches were built from: the tensors' storage and torch's in-place version counters (an
        in-place edit of `powers_of_tau_in_g1` / `inf`, or assigning new tensors, invalidates the shifted table and the folded levels)."""
        p, i = self.powers_of_tau_in_g1, self.inf
        return (p.data_ptr(), p.shape[0], p._version, i.data_ptr(), i._version)

    def _fingerprint(self):
        """Content check for what the stamp cannot see -- writes through raw pointers (a kernel filling the tensors via data_ptr())
        and a new tensor that landed on the same address with the same shape and version: the first and last two points and their
        infinity flags (zkhip_srs_fingerprint: one small launch and one copy, ~20 us against a >= 1 ms commitment)."""
        p, i = self.powers_of_tau_in_g1, self.inf
        n = p.shape[0]
        if n == 0:
            return b""
        out = np.empty(52, dtype=np.uint64)
        ctx = N.Context.get(p.device.index)
        N.check(N.lib().zkhip_srs_fingerprint(ctx.handle, N.ptr(p), N.ptr(i), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)), "srs_fingerprint")
        return out.tobytes()

    def _drop_tables(self):
        """The derived tables go: their addresses are released first (zkhip_table_release) -- the allocator may hand them to a buffer that
        is no table, and the library does not read the header of an address it remembers."""
        for t in (getattr(self, "_table", None), getattr(self, "_level_tables", None)):
            if t is not None:
                try:
                    N.lib().zkhip_table_release(N.Context.get(t.device.index).handle, N.ptr(t))
                except Exception:       # noqa: BLE001 -- interpreter shutdown
                    pass
        self._table = self._folded = self._level_tables = None

    def __del__(self):
        self._drop_tables()

    def invalidate(self):
        """Drops the shifted table and the folded levels (call after writing the SRS through raw pointers; commitments in flight
        keep the table they were started with alive through their PendingCommitment)."""
        self._drop_tables()
        self._cache_stamp = self._cache_print = None

    def _check_caches(self):
        have = getattr(self, "_table", None) is not None or getattr(self, "_folded", None) is not None
        if getattr(self, "_cache_stamp", None) != self._stamp() or (have and getattr(self, "_cache_print", None) != self._fingerprint()):
            self._drop_tables()
            self._cache_stamp = self._stamp()
            self._cache_print = None

    def _caches_built(self):
        if getattr(self, "_cache_print", None) is None:
            self._cache_print = self._fingerprint()

    @property
    def table(self):
        """the shifted-SRS table, or None (never a stale one: see _stamp)"""
        self._check_caches()
        return getattr(self, "_table", None)

    SMALL_SRS = 4096      # zkhip_kzg_commit_table's short path (no sort, no buckets) serves commits of at most this many scalars

    def table_for(self, n_scalars):
        """the table a commitment of n_scalars coefficients should use: a SMALL SRS builds its table on first use (a few ms, a few
        MiB, once per SRS) -- the short commits of plonk-style callers then take two launches instead of the bucket pipeline"""
        t = self.table
        if t is None and 0 < n_scalars <= TrustedSetup.SMALL_SRS and 0 < len(self) <= TrustedSetup.SMALL_SRS:
            t = self.precompute().table
        return t

    def precompute(self):
        """Build (once) the shifted-SRS table 2^(first bit of window w) * point for the digit windows of a scalar (13 at 2^20): commitments then need
        13 instead of 16 bucket additions per point and one bucket reduction (zkhip_srs_precompute); 1.6 GiB at 2^20."""
        self._check_caches()
        if getattr(self, "_table", None) is None:
            import torch
            n = len(self)
            N.lib().zkhip_srs_table_bytes.restype = C.c_size_t
            nbytes = N.lib().zkhip_srs_table_bytes(C.c_size_t(n))
            table = torch.empty(nbytes, dtype=torch.uint8, device=self.powers_of_tau_in_g1.device)
            ctx = N.Context.get(table.device.index)
            N.check(N.lib().zkhip_srs_precompute(ctx.handle, N.ptr(self.powers_of_tau_in_g1), N.ptr(self.inf), C.c_size_t(n),
                                                 N.ptr(table)), "srs_precompute")
            self._table = table
            self._caches_built()
        return self

    def folded(self):
        """The SRS summed over its leading variables, level after level (n - 1 points): what commitments to the
        blown-up quotients of `open` reduce to.  Depends on the SRS only; derived once and kept."""
        self._check_caches()
        if getattr(self, "_folded", None) is None:
            n = len(self)
            xy, inf = TrustedSetup._alloc(n - 1)
            ctx = N.Context.get(self.powers_of_tau_in_g1.device.index)
            N.check(N.lib().zkhip_srs_fold_levels(ctx.handle, N.ptr(self.powers_of_tau_in_g1), N.ptr(self.inf), C.c_size_t(n),
                                                  N.ptr(xy), N.ptr(inf)), "srs_fold_levels")
            self._folded = (xy, inf)
            self._caches_built()
        return self._folded

    @property
    def level_tables(self):
        """the shifted tables of the folded levels, or None (never stale ones: see _stamp)"""
        self._check_caches()
        return getattr(self, "_level_tables", None)

    def precompute_open(self):
        """Build (once) the folded levels and their shifted tables 2^(c w) * S_i[j], c ~ log2 |S_i| (zkhip_srs_level_tables; 1.9 GiB at
        2^20): every `open` against this SRS then spends ~14 instead of ~18 bucket additions per quotient entry, reduces one bucket
        set per round and ends in a host epilogue of ~20 instead of 255 doublings per round."""
        n = len(self)
        if n < 4 or n & (n - 1):
            return self
        fxy, finf = self.folded()
        if getattr(self, "_level_tables", None) is None:
            import torch
            N.lib().zkhip_srs_level_tables_bytes.restype = C.c_size_t
            nbytes = N.lib().zkhip_srs_level_tables_bytes(C.c_size_t(n))
            tables = torch.empty(nbytes, dtype=torch.uint8, device=self.powers_of_tau_in_g1.device)
            ctx = N.Context.get(tables.device.index)
            N.check(N.lib().zkhip_srs_level_tables(ctx.handle, N.ptr(fxy), N.ptr(finf), C.c_size_t(n), N.ptr(tables)), "srs_level_tables")
            self._level_tables = tables
            self._caches_built()
        return self

    @staticmethod
    def _alloc(n):
        import torch
        return (torch.empty((n, 12), dtype=torch.int64, device="cuda"), torch.empty((n,), dtype=torch.uint8, device="cuda"))

    @staticmethod
    def setup(eval_points):
        """TrustedSetup::setup (trusted_setup.rs:15-35): G * eq_i(tau) over the boolean hypercube, MSB first."""
        tau = _fr_host(eval_points)
        nv = tau.shape[0]
        pts, inf = TrustedSetup._alloc(1 << nv)
        ctx = N.Context.get()
        N.check(N.lib().zkhip_srs_multilinear_g1(ctx.handle, tau.ctypes.data_as(C.c_void_p), C.c_uint32(nv),
                                                 N.ptr(pts), N.ptr(inf)), "srs_multilinear")
        return TrustedSetup(pts, inf)


def _commit(points, inf, n_points, scalars, n_scalars, require_equal_len, table=None):
    out = np.empty(12, dtype=np.uint64)
    oinf = C.c_uint8(0)
    ctx = N.Context.get(points.device.index)
    if table is not None:
        st = N.lib().zkhip_kzg_commit_table(ctx.handle, N.ptr(table), N.ptr(inf), C.c_size_t(n_points), N.ptr(scalars),
                                            C.c_size_t(n_scalars), C.c_int(1 if require_equal_len else 0),
                                            out.ctypes.data_as(C.c_void_p), C.byref(oinf))
    else:
        st = N.lib().zkhip_kzg_commit(ctx.handle, N.ptr(points), N.ptr(inf), C.c_size_t(n_points), N.ptr(scalars),
                                      C.c_size_t(n_scalars), C.c_int(1 if require_equal_len else 0),
                                      out.ctypes.data_as(C.c_void_p), C.byref(oinf))
    N.check(st, "The length of powers_of_tau_in_g1 and the length of the evaluations of the polynomial should tally!")
    return G1Affine(out, oinf.value)


class PendingCommitment:
    """A commitment in flight (zkhip_kzg_commit_begin / _end): wait() delivers the G1Affine the synchronous call returns."""

    def __init__(self, ctx, ticket, keep):
        self._ctx, self._ticket, self._keep = ctx, ticket, keep     # `keep`: the tensors the kernels still read

    def wait(self):
        if self._ticket is None:
            raise RuntimeError("commitment already collected")
        out = np.empty(12, dtype=np.uint64)
        oinf = C.c_uint8(0)
        t, self._ticket = self._ticket, None
        N.check(N.lib().zkhip_kzg_commit_end(self._ctx.handle, C.c_uint32(t), out.ctypes.data_as(C.c_void_p), C.byref(oinf)), "commit_end")
        self._keep = None
        return G1Affine(out, oinf.value)

    def __del__(self):
        if getattr(self, "_ticket", None) is not None:
            try:
                N.lib().zkhip_kzg_commit_end(self._ctx.handle, C.c_uint32(self._ticket), None, None)
            except Exception:
                pass


def _commit_begin(points, inf, n_points, scalars, n_scalars, require_equal_len, table=None):
    ctx = N.Context.get(points.device.index)
    ticket = C.c_uint32(0)
    st = N.lib().zkhip_kzg_commit_begin(ctx.handle, None if table is not None else N.ptr(points), N.ptr(table) if table is not None else None,
                                        N.ptr(inf), C.c_size_t(n_points), N.ptr(scalars), C.c_size_t(n_scalars),
                                        C.c_int(1 if require_equal_len else 0), C.byref(ticket))
    N.check(st, "The length of powers_of_tau_in_g1 and the length of the evaluations of the polynomial should tally!")
    return PendingCommitment(ctx, ticket.value, (points, inf, scalars, table))


def commit_batch(points, inf, scalars, offsets):
    """zkhip_kzg_commit_batch: commitments of the slices [offsets[j], offsets[j+1]) of (points, scalars), one pass"""
    nprob = len(offsets) - 1
    xy = np.zeros((max(nprob, 1), 12), dtype=np.uint64)
    oinf = np.zeros(max(nprob, 1), dtype=np.uint8)
    offs = (C.c_size_t * len(offsets))(*[int(o) for o in offsets])
    ctx = N.Context.get(points.device.index)
    N.check(N.lib().zkhip_kzg_commit_batch(ctx.handle, N.ptr(points), N.ptr(inf), N.ptr(scalars), offs, C.c_uint32(nprob),
                                           xy.ctypes.data_as(C.c_void_p), oinf.ctypes.data_as(C.c_void_p)), "commit_batch")
    return [G1Affine(xy[j], oinf[j]) for j in range(nprob)]


class MultilinearKZGProof:
    """kzg/src/multilinear_kzg.rs:17-21: evaluation (Montgomery limbs, uint64 [4]) + one G1 proof per variable"""

    def __init__(self, evaluation, proofs):
        self.evaluation = evaluation
        self.proofs = proofs


class MultilinearKZG:
    @staticmethod
    def commitment(poly, srs):
        """MultilinearKZGInterface::commitment (multilinear_kzg.rs:33-48)"""
        assert isinstance(poly, Multilinear)
        return _commit(srs.powers_of_tau_in_g1, srs.inf, len(srs), poly.evaluations, len(poly), True, srs.table_for(len(poly)))

    @staticmethod
    def commitment_begin(poly, srs):
        """The same commitment, in flight: -> PendingCommitment (at most three at a time); .wait() yields the G1Affine."""
        assert isinstance(poly, Multilinear)
        return _commit_begin(srs.powers_of_tau_in_g1, srs.inf, len(srs), poly.evaluations, len(poly), True, srs.table)

    @staticmethod
    def open(poly, evaluation_points, srs, cache_folded_srs=True):
        """MultilinearKZGInterface::open (multilinear_kzg.rs:50-88)"""
        assert isinstance(poly, Multilinear)
        pts = _fr_host(evaluation_points)
        nv = pts.shape[0]
        ev = np.empty(4, dtype=np.uint64)
        pxy = np.zeros((max(nv, 1), 12), dtype=np.uint64)
        pinf = np.zeros(max(nv, 1), dtype=np.uint8)
        ctx = N.Context.get(poly.evaluations.device.index)
        n = len(poly)
        tables = None
        if cache_folded_srs and len(srs) == n and n >= 4 and n & (n - 1) == 0:
            fxy, finf = srs.folded()
            fxy_p, finf_p = N.ptr(fxy), N.ptr(finf)
            tables = getattr(srs, "_level_tables", None)       # present after srs.precompute_open(); folded() has just run the cache guard
            if tables is None and n <= TrustedSetup.SMALL_SRS:  # a small SRS builds them on first use (a few ms): its openings take the
                tables = srs.precompute_open()._level_tables    # short path, two launches for all rounds (zkhip_kzg_open_tables)
        else:
            fxy_p = finf_p = None
        st = N.lib().zkhip_kzg_open_tables(ctx.handle, N.ptr(poly.evaluations), C.c_size_t(n), pts.ctypes.data_as(C.c_void_p),
                                           C.c_size_t(nv), N.ptr(srs.powers_of_tau_in_g1), N.ptr(srs.inf), C.c_size_t(len(srs)),
                                           fxy_p, finf_p, N.ptr(tables) if tables is not None else None,
                                           ev.ctypes.data_as(C.c_void_p), pxy.ctypes.data_as(C.c_void_p),
                                           pinf.ctypes.data_as(C.c_void_p))
        N.check(st, "open: points / SRS length must match the polynomial (and n_vars >= 2)")
        return MultilinearKZGProof(ev, [G1Affine(pxy[i], pinf[i]) for i in range(nv)])


class UnivariateKZGProof:
    """kzg/src/univariate_kzg.rs:11-15: evaluation (Montgomery limbs uint64 [4]) + the quotient's commitment"""

    def __init__(self, evaluation, proof):
        self.evaluation = evaluation
        self.proof = proof


class UnivariateKZG:
    @staticmethod
    def open(poly, evaluation_point, srs):
        """UnivariateKZGInterface::open (univariate_kzg.rs:60-81)"""
        assert isinstance(poly, DenseUnivariatePolynomial)
        z = _fr_host(evaluation_point).reshape(4)
        ev, xy, inf = np.empty(4, dtype=np.uint64), np.empty(12, dtype=np.uint64), C.c_uint8(0)
        ctx = N.Context.get(srs.powers_of_tau_in_g1.device.index)
        st = N.lib().zkhip_univariate_kzg_open(ctx.handle, N.ptr(poly.coefficients) if len(poly) else None, C.c_size_t(len(poly)),
                                               z.ctypes.data_as(C.c_void_p), N.ptr(srs.powers_of_tau_in_g1), N.ptr(srs.inf),
                                               C.c_size_t(len(srs)), ev.ctypes.data_as(C.c_void_p), xy.ctypes.data_as(C.c_void_p),
                                               C.byref(inf))
        N.check(st, "univariate open")
        return UnivariateKZGProof(ev, G1Affine(xy, inf.value))

    @staticmethod
    def generate_srs(tau, max_degree):
        """UnivariateKZGInterface::generate_srs (univariate_kzg.rs:18-35), G1 powers"""
        t = _fr_host(tau)
        pts, inf = TrustedSetup._alloc(max_degree + 1)
        ctx = N.Context.get()
        N.check(N.lib().zkhip_srs_univariate_g1(ctx.handle, t.ctypes.data_as(C.c_void_p), C.c_size_t(max_degree),
                                                N.ptr(pts), N.ptr(inf)), "srs_univariate")
        return TrustedSetup(pts, inf)

    @staticmethod
    def commitment(poly, srs):
        """UnivariateKZGInterface::commitment (univariate_kzg.rs:37-58): no length assert; a polynomial longer
        than the SRS indexes out of bounds (IndexError), exactly what the unregistered bench would hit."""
        assert isinstance(poly, DenseUnivariatePolynomial)
        return _commit(srs.powers_of_tau_in_g1, srs.inf, len(srs), poly.coefficients, len(poly), False, srs.table_for(len(poly)))

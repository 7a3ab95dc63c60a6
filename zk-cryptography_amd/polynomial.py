"""polynomial::Multilinear (evaluation form) on the GPU.

Mirrors polynomial/src/multilinear/evaluation_form.rs and MultilinearTrait
(polynomial/src/interface.rs:9-13): same method names, argument meaning and error
behaviour (the reference's assert!/panic! surface as AssertionError).  The evaluation
table lives in HBM as an int64 [n, 4] torch tensor holding arkworks' Montgomery limbs.
"""
import ctypes as C

import numpy as np

from zk_cryptography_amd import _native as N


def _to_device(evals, device=None):
    import torch
    if isinstance(evals, torch.Tensor):
        t = evals
        if t.dtype != torch.int64 or t.dim() != 2 or t.shape[1] != 4:
            raise AssertionError("evaluations tensor must be int64 [n, 4]")
        if not t.is_cuda:
            t = t.cuda(device)
        return t.contiguous()
    a = np.ascontiguousarray(evals, dtype=np.uint64)
    if a.ndim != 2 or a.shape[1] != 4:
        raise AssertionError("evaluations must be uint64 [n, 4] Montgomery limbs")
    return torch.from_numpy(a.view(np.int64)).cuda(device)


def _fr_host(x):
    a = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
    return a


class Multilinear:
    def __init__(self, evaluations, device=None):
        """Multilinear::new (evaluation_form.rs:12-26): the number of evaluations must be a power of 2."""
        t = _to_device(evaluations, device)
        n = t.shape[0]
        if n == 0 or (n & (n - 1)):
            raise AssertionError("Number of evaluations must be a power of 2")
        self.evaluations = t
        self.n_vars = n.bit_length() - 1

    # -- helpers -------------------------------------------------------------------------
    @property
    def _ctx(self):
        return N.Context.get(self.evaluations.device.index)

    def __len__(self):
        return self.evaluations.shape[0]

    def to_numpy(self):
        return self.evaluations.cpu().numpy().view(np.uint64)

    def _new_like(self, n):
        import torch
        return torch.empty((n, 4), dtype=torch.int64, device=self.evaluations.device)

    @classmethod
    def _wrap(cls, t, n_vars=None):
        m = cls.__new__(cls)
        m.evaluations = t
        m.n_vars = t.shape[0].bit_length() - 1 if n_vars is None else n_vars
        return m

    # -- MultilinearTrait ------------------------------------------------------------------
    def partial_evaluation(self, eval_point, variable_index):
        """evaluation_form.rs:123-141"""
        n = len(self)
        out = self._new_like(max(n // 2, 1))
        r = _fr_host(eval_point)
        st = N.lib().zkhip_mle_partial_evaluation(self._ctx.handle, N.ptr(self.evaluations), C.c_size_t(n),
                                                  r.ctypes.data_as(C.c_void_p), None, C.c_uint32(variable_index),
                                                  N.ptr(out))
        N.check(st, "partial_evaluation")
        if n >= 2 and (1 << variable_index) >= n:
            # the reference's pair list is empty here (utils.rs:37-50): an EMPTY table with n_vars - 1 (struct literal, :137-140)
            return Multilinear._wrap(out[:0], self.n_vars - 1)
        return Multilinear._wrap(out[: n // 2])

    def partial_evaluations(self, points, variable_indices):
        """evaluation_form.rs:143-159"""
        pts = _fr_host(points)
        if pts.shape[0] != len(variable_indices):
            raise AssertionError("The length of evaluation_points and variable_indices should be the same: %d, %d"
                                 % (pts.shape[0], len(variable_indices)))
        n = len(self)
        k = pts.shape[0]
        if k > self.n_vars:
            raise AssertionError("more points than variables")
        out = self._new_like(n >> k)
        idx = (C.c_uint32 * max(k, 1))(*variable_indices)
        st = N.lib().zkhip_mle_partial_evaluations(self._ctx.handle, N.ptr(self.evaluations), C.c_size_t(n),
                                                   pts.ctypes.data_as(C.c_void_p), idx, C.c_size_t(k), N.ptr(out))
        N.check(st, "partial_evaluations")
        return Multilinear._wrap(out)

    def evaluation(self, evaluation_points):
        """evaluation_form.rs:162-175 -> uint64[4] (Montgomery)"""
        pts = _fr_host(evaluation_points)
        out = np.empty(4, dtype=np.uint64)
        st = N.lib().zkhip_mle_evaluation(self._ctx.handle, N.ptr(self.evaluations), C.c_size_t(len(self)),
                                          pts.ctypes.data_as(C.c_void_p), C.c_size_t(pts.shape[0]),
                                          out.ctypes.data_as(C.c_void_p))
        N.check(st, "Number of evaluation points must match the number of variables")
        return out

    # -- inherent methods ------------------------------------------------------------------
    def _half_sums(self):
        out = np.empty((3, 4), dtype=np.uint64)
        st = N.lib().zkhip_mle_half_sums(self._ctx.handle, N.ptr(self.evaluations), C.c_size_t(len(self)),
                                         out.ctypes.data_as(C.c_void_p))
        N.check(st, "half_sums")
        return out

    def split_poly_into_two_and_sum_each_part(self):
        """evaluation_form.rs:68-74 -> Multilinear of 2 evaluations"""
        return Multilinear(self._half_sums()[:2], self.evaluations.device)

    def sum_over_the_boolean_hypercube(self):
        """evaluation_form.rs:80-84"""
        if len(self) == 1:
            return self.to_numpy()[0]
        return self._half_sums()[2]

    def _distinct(self, rhs, fn):
        out = self._new_like(len(self) * len(rhs))
        st = fn(self._ctx.handle, N.ptr(self.evaluations), C.c_size_t(len(self)), N.ptr(rhs.evaluations),
                C.c_size_t(len(rhs)), N.ptr(out))
        N.check(st, "distinct")
        return Multilinear._wrap(out)

    def add_to_front(self, variable_length):
        """evaluation_form.rs:86-96: the table repeated 2 * 2^variable_length times"""
        out = self._new_like(len(self) * 2 * (1 << variable_length))
        N.check(N.lib().zkhip_mle_add_to_front(self._ctx.handle, N.ptr(self.evaluations), C.c_size_t(len(self)),
                                               C.c_uint32(variable_length), N.ptr(out)), "add_to_front")
        return Multilinear._wrap(out)

    def add_to_back(self, variable_length):
        """evaluation_form.rs:98-110: every entry repeated 2^variable_length times"""
        out = self._new_like(len(self) << variable_length)
        N.check(N.lib().zkhip_mle_add_to_back(self._ctx.handle, N.ptr(self.evaluations), C.c_size_t(len(self)),
                                              C.c_uint32(variable_length), N.ptr(out)), "add_to_back")
        return Multilinear._wrap(out)

    @staticmethod
    def duplicate_evaluation(value):
        """evaluation_form.rs:112-119: value ++ value"""
        v = _fr_host(value)
        return Multilinear(np.concatenate([v, v]))

    @staticmethod
    def additive_identity(num_vars):
        """evaluation_form.rs:64-66"""
        return Multilinear(np.zeros((1 << num_vars, 4), dtype=np.uint64))

    def add_distinct(self, rhs):
        """evaluation_form.rs:28-39"""
        return self._distinct(rhs, N.lib().zkhip_mle_add_distinct)

    def mul_distinct(self, rhs):
        """evaluation_form.rs:41-52"""
        return self._distinct(rhs, N.lib().zkhip_mle_mul_distinct)

    def to_bytes(self):
        """evaluation_form.rs:54-62"""
        import torch
        out = torch.empty(32 * len(self), dtype=torch.uint8, device=self.evaluations.device)
        N.check(N.lib().zkhip_mle_to_bytes(self._ctx.handle, N.ptr(self.evaluations), C.c_size_t(len(self)),
                                           N.ptr(out)), "to_bytes")
        return out.cpu().numpy().tobytes()

    def _elementwise(self, op, other=None, scalar=None):
        out = self._new_like(len(self))
        sc = _fr_host(scalar) if scalar is not None else None
        st = N.lib().zkhip_mle_elementwise(self._ctx.handle, C.c_int(op), N.ptr(self.evaluations),
                                           N.ptr(other.evaluations) if other is not None else None,
                                           sc.ctypes.data_as(C.c_void_p) if sc is not None else None,
                                           C.c_size_t(len(self)), C.c_size_t(len(other) if other is not None else 0), N.ptr(out))
        N.check(st, "elementwise")     # a shorter rhs is the reference's index panic (evaluation_form.rs:185) -> IndexError
        return Multilinear._wrap(out)

    def __add__(self, rhs):   # evaluation_form.rs:178-194
        return self._elementwise(0, other=rhs)

    def __sub__(self, rhs):   # :208-224
        return self._elementwise(1, other=rhs)

    def __mul__(self, scalar):   # Mul<F> :235-251
        return self._elementwise(2, scalar=scalar)

    def __eq__(self, other):
        import torch
        return isinstance(other, Multilinear) and self.n_vars == other.n_vars and \
            bool(torch.equal(self.evaluations, other.evaluations))

"""ctypes binding of the CPU ORACLE (oracle/libzkoracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

Array conventions (same as the product C-ABI):
  Fr table  -> numpy uint64 [n, 4]  little-endian limbs, Montgomery form
  G1 affine -> numpy uint64 [n, 13] (x[6], y[6], inf)   Montgomery Fq limbs
  G1 jac    -> numpy uint64 [n, 18] (X[6], Y[6], Z[6])
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libzkoracle.so")

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Q_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
FR_R = pow(2, 256, R_MOD)
FQ_R = pow(2, 384, Q_MOD)
SPARSE_MAX = 16


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libzkoracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.ora_sparse_to_bytes.restype = C.c_size_t
        _lib.ora_dense_mul.restype = C.c_size_t
    return _lib


class Sparse(C.Structure):
    _fields_ = [("coeff", C.c_uint64 * (4 * SPARSE_MAX)), ("pow", C.c_uint64 * (4 * SPARSE_MAX)), ("len", C.c_size_t)]

    def monomials(self):
        """[(coeff_int, pow_int)] in canonical integers."""
        out = []
        for k in range(self.len):
            c = limbs_to_int(list(self.coeff[4 * k:4 * k + 4])) * pow(FR_R, -1, R_MOD) % R_MOD
            p = limbs_to_int(list(self.pow[4 * k:4 * k + 4])) * pow(FR_R, -1, R_MOD) % R_MOD
            out.append((c, p))
        return out


# ---- int <-> limb helpers (pure python ints: independent of the C code) ------
def limbs_to_int(limbs):
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def int_to_limbs(v, n):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def fr_from_ints(vals):
    """canonical python ints (any sign/size) -> Montgomery uint64 [n,4]"""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = int_to_limbs((int(v) % R_MOD) * FR_R % R_MOD, 4)
    return out


def fr_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    rinv = pow(FR_R, -1, R_MOD)
    return [limbs_to_int(row) * rinv % R_MOD for row in arr]


def fq_from_ints(vals):
    out = np.empty((len(vals), 6), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = int_to_limbs((int(v) % Q_MOD) * FQ_R % Q_MOD, 6)
    return out


def fq_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 6)
    rinv = pow(FQ_R, -1, Q_MOD)
    return [limbs_to_int(row) * rinv % Q_MOD for row in arr]


def random_fr(n, seed):
    """n uniform field elements in Montgomery form (numpy PCG64; any 4 limbs < r is a valid Montgomery residue)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)   # < 2^254 < r: uniform on [0, 2^254), all valid residues
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _fr(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4
    return a


# ---- Fr scalar ops ------------------------------------------------------------
def _bin(fn, a, b):
    a, b = _fr(a).reshape(4), _fr(b).reshape(4)
    o = np.empty(4, dtype=np.uint64)
    getattr(lib(), fn)(_p(o), _p(a), _p(b))
    return o


def fr_add(a, b): return _bin("ora_fr_add", a, b)
def fr_sub(a, b): return _bin("ora_fr_sub", a, b)
def fr_mul(a, b): return _bin("ora_fr_mul", a, b)


def fr_inv(a):
    a = _fr(a).reshape(4)
    o = np.empty(4, dtype=np.uint64)
    assert lib().ora_fr_inv(_p(o), _p(a)) == 1
    return o


def fr_to_bytes_be(a):
    a = _fr(a).reshape(4)
    o = np.empty(32, dtype=np.uint8)
    lib().ora_fr_to_bytes_be(_p(o), _p(a))
    return o.tobytes()


def fr_from_be_bytes_mod_order(b):
    buf = np.frombuffer(bytes(b), dtype=np.uint8).copy()
    o = np.empty(4, dtype=np.uint64)
    lib().ora_fr_from_be_bytes_mod_order(_p(o), _p(buf), C.c_size_t(len(buf)))
    return o


def fr_get_root_of_unity(n):
    o = np.empty(4, dtype=np.uint64)
    assert lib().ora_fr_get_root_of_unity(_p(o), C.c_uint64(n)) == 1
    return o


def fq_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(6)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(6)
    o = np.empty(6, dtype=np.uint64)
    lib().ora_fq_mul(_p(o), _p(a), _p(b))
    return o


# ---- SHA-256 / transcript -------------------------------------------------------
class Transcript:
    """fiat_shamir.rs:10-40"""

    def __init__(self):
        self._t = (C.c_uint8 * 128)()
        lib().ora_transcript_new(self._t)

    def commit(self, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        lib().ora_transcript_commit(self._t, _p(buf), C.c_size_t(len(buf)))

    def challenge(self):
        o = np.empty(32, dtype=np.uint8)
        lib().ora_transcript_challenge(self._t, _p(o))
        return o.tobytes()

    def evaluate_challenge_into_field(self):
        o = np.empty(4, dtype=np.uint64)
        lib().ora_transcript_challenge_fr(self._t, _p(o))
        return o


def sha256(data):
    st = (C.c_uint8 * 128)()
    lib().ora_sha256_init(st)
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    lib().ora_sha256_update(st, _p(buf), C.c_size_t(len(buf)))
    o = np.empty(32, dtype=np.uint8)
    lib().ora_sha256_final(st, _p(o))
    return o.tobytes()


# ---- Multilinear ------------------------------------------------------------------
def mle_partial_evaluation(evals, r, var_index, mt=False):
    evals, r = _fr(evals), _fr(r).reshape(4)
    n = evals.shape[0]
    out = np.empty((max(n // 2, 1), 4), dtype=np.uint64)
    fn = lib().ora_mle_partial_evaluation_mt if mt else lib().ora_mle_partial_evaluation
    rc = fn(_p(out), _p(evals), C.c_size_t(n), _p(r), C.c_size_t(var_index))
    if rc:
        raise AssertionError("n must be even" if rc == -1 else "variable_index must be less than n/2")
    return out[: n // 2]


def mle_partial_evaluations(evals, pts, var_indices):
    evals, pts = _fr(evals), _fr(pts).reshape(-1, 4)
    if len(pts) != len(var_indices):
        raise AssertionError("The length of evaluation_points and variable_indices should be the same")
    n = evals.shape[0]
    out = np.empty((n, 4), dtype=np.uint64)
    out_n = C.c_size_t(0)
    idx = (C.c_size_t * max(len(var_indices), 1))(*var_indices)
    rc = lib().ora_mle_partial_evaluations(_p(out), C.byref(out_n), _p(evals), C.c_size_t(n), _p(pts), idx,
                                           C.c_size_t(len(var_indices)))
    if rc:
        raise AssertionError("partial_evaluations shape error %d" % rc)
    return out[: out_n.value].copy()


def mle_evaluation(evals, pts):
    evals, pts = _fr(evals), _fr(pts).reshape(-1, 4)
    o = np.empty(4, dtype=np.uint64)
    rc = lib().ora_mle_evaluation(_p(o), _p(evals), C.c_size_t(evals.shape[0]), _p(pts), C.c_size_t(pts.shape[0]))
    if rc:
        raise AssertionError("Number of evaluation points must match the number of variables")
    return o


def mle_half_sums(evals):
    evals = _fr(evals)
    o = np.empty((2, 4), dtype=np.uint64)
    lib().ora_mle_half_sums(_p(o), _p(evals), C.c_size_t(evals.shape[0]))
    return o


def mle_sum(evals):
    evals = _fr(evals)
    o = np.empty(4, dtype=np.uint64)
    lib().ora_mle_sum(_p(o), _p(evals), C.c_size_t(evals.shape[0]))
    return o


def mle_add_to_front(evals, variable_length):
    evals = _fr(evals).reshape(-1, 4)
    o = np.empty((evals.shape[0] * 2 * (1 << variable_length), 4), dtype=np.uint64)
    lib().ora_mle_add_to_front(_p(o), _p(evals), C.c_size_t(evals.shape[0]), C.c_size_t(variable_length))
    return o


def mle_add_to_back(evals, variable_length):
    evals = _fr(evals).reshape(-1, 4)
    o = np.empty((evals.shape[0] << variable_length, 4), dtype=np.uint64)
    lib().ora_mle_add_to_back(_p(o), _p(evals), C.c_size_t(evals.shape[0]), C.c_size_t(variable_length))
    return o


def mle_add_distinct(a, b):
    a, b = _fr(a), _fr(b)
    o = np.empty((a.shape[0] * b.shape[0], 4), dtype=np.uint64)
    lib().ora_mle_add_distinct(_p(o), _p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0]))
    return o


def mle_mul_distinct(a, b):
    a, b = _fr(a), _fr(b)
    o = np.empty((a.shape[0] * b.shape[0], 4), dtype=np.uint64)
    lib().ora_mle_mul_distinct(_p(o), _p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0]))
    return o


def mle_to_bytes(evals):
    evals = _fr(evals).reshape(-1, 4)
    o = np.empty(32 * evals.shape[0], dtype=np.uint8)
    lib().ora_mle_to_bytes(_p(o), _p(evals), C.c_size_t(evals.shape[0]))
    return o.tobytes()


# ---- Sparse univariate ---------------------------------------------------------------
def sparse_interpolation(xs, ys):
    xs, ys = _fr(xs).reshape(-1, 4), _fr(ys).reshape(-1, 4)
    s = Sparse()
    assert lib().ora_sparse_interpolation(C.byref(s), _p(xs), _p(ys), C.c_size_t(xs.shape[0])) == 0
    return s


def sparse_add(a, b):
    s = Sparse()
    lib().ora_sparse_add(C.byref(s), C.byref(a), C.byref(b))
    return s


def sparse_evaluate(p, x):
    x = _fr(x).reshape(4)
    o = np.empty(4, dtype=np.uint64)
    lib().ora_sparse_evaluate(_p(o), C.byref(p), _p(x))
    return o


def sparse_to_bytes(p):
    o = np.empty(64 * SPARSE_MAX, dtype=np.uint8)
    n = lib().ora_sparse_to_bytes(_p(o), C.byref(p))
    return o[:n].tobytes()


# ---- Sumcheck ----------------------------------------------------------------------------
def _nvars(n):
    k = n.bit_length() - 1
    if (1 << k) != n:
        raise AssertionError("Number of evaluations must be a power of 2")
    return k


def sumcheck_prove(evals):
    """-> (sum[4], round_polys[n_vars,2,4], challenges[n_vars,4])"""
    evals = _fr(evals)
    n = evals.shape[0]
    nv = _nvars(n)
    s = np.empty(4, dtype=np.uint64)
    rp = np.empty((max(nv, 1), 2, 4), dtype=np.uint64)
    ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
    assert lib().ora_sumcheck_prove(_p(evals), C.c_size_t(n), _p(s), _p(rp), _p(ch)) == 0
    return s, rp[:nv], ch[:nv]


def sumcheck_verify(evals, s, round_polys):
    evals, s, rp = _fr(evals), _fr(s).reshape(4), _fr(round_polys)
    return bool(lib().ora_sumcheck_verify(_p(evals), C.c_size_t(evals.shape[0]), _p(s), _p(rp)))


def composed_sum(tables):
    tables = _fr(tables)
    k, n = tables.shape[0], tables.shape[1]
    o = np.empty(4, dtype=np.uint64)
    lib().ora_composed_sum(_p(o), _p(tables), C.c_size_t(k), C.c_size_t(n))
    return o


def composed_prove(tables):
    """tables [K, n, 4] -> (round_polys [n_vars, K+1, 4], challenges [n_vars, 4])"""
    tables = _fr(tables)
    k, n = tables.shape[0], tables.shape[1]
    nv = _nvars(n)
    rp = np.empty((max(nv, 1), k + 1, 4), dtype=np.uint64)
    ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
    assert lib().ora_composed_prove(_p(tables), C.c_size_t(k), C.c_size_t(n), _p(rp), _p(ch)) == 0
    return rp[:nv], ch[:nv]


def composed_verify(tables, s, round_polys):
    tables, s, rp = _fr(tables), _fr(s).reshape(4), _fr(round_polys)
    return bool(lib().ora_composed_verify(_p(tables), C.c_size_t(tables.shape[0]), C.c_size_t(tables.shape[1]),
                                          _p(s), _p(rp)))


def multi_composed_sum(tables, term_sizes):
    tables = _fr(tables)
    ts = (C.c_size_t * len(term_sizes))(*term_sizes)
    o = np.empty(4, dtype=np.uint64)
    lib().ora_multi_composed_sum(_p(o), _p(tables), ts, C.c_size_t(len(term_sizes)), C.c_size_t(tables.shape[1]))
    return o


def multi_composed_prove(tables, term_sizes, s, partial):
    """tables [sum(term_sizes), n, 4] -> (list[Sparse] per round, challenges [n_vars,4])"""
    tables, s = _fr(tables), _fr(s).reshape(4)
    n = tables.shape[1]
    nv = _nvars(n)
    ts = (C.c_size_t * len(term_sizes))(*term_sizes)
    rps = (Sparse * max(nv, 1))()
    ch = np.empty((max(nv, 1), 4), dtype=np.uint64)
    assert lib().ora_multi_composed_prove(_p(tables), ts, C.c_size_t(len(term_sizes)), C.c_size_t(n), _p(s),
                                          C.c_int(1 if partial else 0), rps, _p(ch)) == 0
    return [rps[i] for i in range(nv)], ch[:nv]


def multi_composed_verify(tables, term_sizes, s, round_polys):
    tables, s = _fr(tables), _fr(s).reshape(4)
    ts = (C.c_size_t * len(term_sizes))(*term_sizes)
    arr = (Sparse * max(len(round_polys), 1))(*round_polys)
    return lib().ora_multi_composed_verify(_p(tables), ts, C.c_size_t(len(term_sizes)), C.c_size_t(tables.shape[1]),
                                           _p(s), arr, C.c_size_t(len(round_polys)))


def multi_composed_proof_bytes(round_polys):
    """ComposedSumcheckProof::to_bytes (multi_composed_sumcheck.rs:24-31)"""
    return b"".join(sparse_to_bytes(p) for p in round_polys)


# ---- G1 / KZG ---------------------------------------------------------------------------------
def g1_generator():
    o = np.empty(18, dtype=np.uint64)
    lib().ora_g1_generator(_p(o))
    return o


def g1_identity():
    o = np.empty(18, dtype=np.uint64)
    lib().ora_g1_identity(_p(o))
    return o


def g1_add(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(18)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(18)
    o = np.empty(18, dtype=np.uint64)
    lib().ora_g1_add(_p(o), _p(a), _p(b))
    return o


def g1_mul_int(base, k):
    base = np.ascontiguousarray(base, dtype=np.uint64).reshape(18)
    sc = np.array(int_to_limbs(int(k) % R_MOD, 4), dtype=np.uint64)
    o = np.empty(18, dtype=np.uint64)
    lib().ora_g1_mul_bigint(_p(o), _p(base), _p(sc), C.c_size_t(4))
    return o


def g1_to_affine(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(18)
    o = np.empty(13, dtype=np.uint64)
    lib().ora_g1_to_affine(_p(o), _p(a))
    return o


def g1_batch_to_affine(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 18)
    o = np.empty((a.shape[0], 13), dtype=np.uint64)
    lib().ora_g1_batch_to_affine(_p(o), _p(a), C.c_size_t(a.shape[0]))
    return o


def g1_is_on_curve(aff):
    aff = np.ascontiguousarray(aff, dtype=np.uint64).reshape(13)
    return bool(lib().ora_g1_is_on_curve(_p(aff)))


def g1_affine_ints(aff):
    """affine [13] -> (x_int, y_int, inf)"""
    aff = np.asarray(aff, dtype=np.uint64).reshape(13)
    x, = fq_to_ints(aff[0:6])
    y, = fq_to_ints(aff[6:12])
    return x, y, bool(aff[12])


def kzg_eq_points(tau):
    tau = _fr(tau).reshape(-1, 4)
    nv = tau.shape[0]
    o = np.empty((1 << nv, 4), dtype=np.uint64)
    lib().ora_kzg_eq_points(_p(o), _p(tau), C.c_size_t(nv))
    return o


def kzg_multilinear_srs_g1(tau):
    tau = _fr(tau).reshape(-1, 4)
    nv = tau.shape[0]
    o = np.empty((1 << nv, 18), dtype=np.uint64)
    lib().ora_kzg_multilinear_srs_g1(_p(o), _p(tau), C.c_size_t(nv))
    return o


def kzg_univariate_srs_g1(tau, max_degree):
    tau = _fr(tau).reshape(4)
    o = np.empty((max_degree + 1, 18), dtype=np.uint64)
    lib().ora_kzg_univariate_srs_g1(_p(o), _p(tau), C.c_size_t(max_degree))
    return o


def kzg_commitment(coeffs, srs_jac, require_equal_len):
    coeffs = _fr(coeffs).reshape(-1, 4)
    srs = np.ascontiguousarray(srs_jac, dtype=np.uint64).reshape(-1, 18)
    o = np.empty(18, dtype=np.uint64)
    rc = lib().ora_kzg_commitment(_p(o), _p(coeffs), C.c_size_t(coeffs.shape[0]), _p(srs), C.c_size_t(srs.shape[0]),
                                  C.c_int(1 if require_equal_len else 0))
    if rc == -1:
        raise AssertionError("The length of powers_of_tau_in_g1 and the length of the evaluations of the polynomial should tally!")
    if rc == -2:
        raise IndexError("index out of bounds: srs shorter than polynomial")
    return o


def kzg_open(evals, points, srs_jac):
    """MultilinearKZG::open (multilinear_kzg.rs:50-88), naive -> (evaluation [4], proofs [n_vars, 18] Jacobian)"""
    evals = _fr(evals).reshape(-1, 4)
    points = _fr(points).reshape(-1, 4)
    srs = np.ascontiguousarray(srs_jac, dtype=np.uint64).reshape(-1, 18)
    ev = np.empty(4, dtype=np.uint64)
    proofs = np.empty((max(points.shape[0], 1), 18), dtype=np.uint64)
    rc = lib().ora_kzg_open(_p(ev), _p(proofs), _p(evals), C.c_size_t(evals.shape[0]), _p(points), C.c_size_t(points.shape[0]),
                            _p(srs), C.c_size_t(srs.shape[0]))
    if rc == -1:
        raise AssertionError("shape mismatch (points / srs length)")
    if rc == -2:
        raise OverflowError("attempt to subtract with overflow (variable_index - 1, multilinear_kzg.rs:73)")
    if rc == -3:
        raise RuntimeError("Evaluation and final remainder mismatch!")
    return ev, proofs[: points.shape[0]]


def univariate_kzg_open(coeffs, z, srs_jac):
    """UnivariateKZG::open (univariate_kzg.rs:60-81) -> (evaluation [4], proof [18] Jacobian)"""
    coeffs, z = _fr(coeffs).reshape(-1, 4), _fr(z).reshape(4)
    srs = np.ascontiguousarray(srs_jac, dtype=np.uint64).reshape(-1, 18)
    ev, proof = np.empty(4, dtype=np.uint64), np.empty(18, dtype=np.uint64)
    rc = lib().ora_univariate_kzg_open(_p(ev), _p(proof), _p(coeffs), C.c_size_t(coeffs.shape[0]), _p(z), _p(srs), C.c_size_t(srs.shape[0]))
    if rc == -2:
        raise IndexError("index out of bounds: srs shorter than the quotient")
    assert rc == 0
    return ev, proof


def dense_divide(a, b):
    """divide_with_q_and_r (dense_univariate.rs:88-124) -> (quotient, remainder)"""
    a, b = _fr(a).reshape(-1, 4), _fr(b).reshape(-1, 4)
    q, r = np.empty((max(a.shape[0], 1), 4), dtype=np.uint64), np.empty((max(a.shape[0], 1), 4), dtype=np.uint64)
    nq, nr = C.c_size_t(0), C.c_size_t(0)
    rc = lib().ora_dense_divide(_p(q), C.byref(nq), _p(r), C.byref(nr), _p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0]))
    if rc != 0:
        raise ZeroDivisionError("Dividing by zero polynomial")
    return q[: nq.value], r[: nr.value]


def msm_pippenger(scalars, pts_affine):
    scalars = _fr(scalars).reshape(-1, 4)
    pts = np.ascontiguousarray(pts_affine, dtype=np.uint64).reshape(-1, 13)
    assert scalars.shape[0] == pts.shape[0]
    o = np.empty(18, dtype=np.uint64)
    lib().ora_msm_pippenger(_p(o), _p(scalars), _p(pts), C.c_size_t(pts.shape[0]))
    return o


# ---- circuit + GKR (gkr.c) ---------------------------------------------------------------------
GKR_MAX_LAYERS, GKR_MAX_ROUNDS = 24, 48


class GkrProof(C.Structure):
    _fields_ = [("n_proofs", C.c_size_t), ("sums", C.c_uint64 * (4 * GKR_MAX_LAYERS)), ("n_rounds", C.c_size_t * GKR_MAX_LAYERS),
                ("round_polys", (Sparse * GKR_MAX_ROUNDS) * GKR_MAX_LAYERS), ("wb", C.c_uint64 * (4 * GKR_MAX_LAYERS)),
                ("wc", C.c_uint64 * (4 * GKR_MAX_LAYERS)), ("w0", C.c_uint64 * 8),
                ("challenges", (C.c_uint64 * (4 * GKR_MAX_ROUNDS)) * GKR_MAX_LAYERS)]

    def layer_challenges(self, k):
        """the challenges prove_partial returned for sumcheck proof k: uint64 [n_rounds, 4]"""
        return np.array(self.challenges[k][0:4 * self.n_rounds[k]], dtype=np.uint64).reshape(-1, 4)

    def layer(self, k):
        """(sum [4], [Sparse per round], wb [4], wc [4]) of sumcheck proof k"""
        return (np.array(self.sums[4 * k:4 * k + 4], dtype=np.uint64), [self.round_polys[k][r] for r in range(self.n_rounds[k])],
                np.array(self.wb[4 * k:4 * k + 4], dtype=np.uint64), np.array(self.wc[4 * k:4 * k + 4], dtype=np.uint64))


    def fields(self):
        """everything a GKRProof holds, as plain python values (for == between two provers)"""
        out = [tuple(self.w0[0:8])]
        for k in range(self.n_proofs):
            s, rps, wb, wc = self.layer(k)
            out.append((tuple(int(v) for v in s), tuple(int(v) for v in wb), tuple(int(v) for v in wc), self.layer_challenges(k).tobytes(),
                        tuple((rp.len, tuple(rp.coeff[0:4 * rp.len]), tuple(rp.pow[0:4 * rp.len])) for rp in rps)))
        return out


def _circuit_args(layers):
    """layers: [[(gate_type, in0, in1), ...], ...] with gate_type 'add' | 'mul', layer 0 = output"""
    n_gates = (C.c_size_t * len(layers))(*[len(l) for l in layers])
    flat = [g for l in layers for g in l]
    gt = np.array([0 if g[0] == "add" else 1 for g in flat], dtype=np.uint8)
    i0 = np.array([g[1] for g in flat], dtype=np.uint32)
    i1 = np.array([g[2] for g in flat], dtype=np.uint32)
    return C.c_size_t(len(layers)), n_gates, gt, i0, i1


def gkr_mle_size(layer_index):
    lib().ora_gkr_mle_size.restype = C.c_size_t
    return lib().ora_gkr_mle_size(C.c_size_t(layer_index))


def circuit_evaluation(layers, inp):
    """Circuit::evaluation (circuit.rs:31-57) -> list of uint64 [len, 4] arrays, output layer first, input last"""
    inp = _fr(inp).reshape(-1, 4)
    nl, ng, gt, i0, i1 = _circuit_args(layers)
    total = inp.shape[0] + sum(len(l) for l in layers)
    out = np.empty((total, 4), dtype=np.uint64)
    lens = (C.c_size_t * (len(layers) + 1))()
    rc = lib().ora_circuit_evaluation(nl, ng, _p(gt), _p(i0), _p(i1), _p(inp), C.c_size_t(inp.shape[0]), _p(out), lens)
    if rc != 0:
        raise IndexError("gate input out of range")
    res, off = [], 0
    for k in range(len(layers) + 1):
        res.append(out[off:off + lens[k]].copy())
        off += lens[k]
    return res


def circuit_add_mult_mle(layers, layer_index):
    nl, ng, gt, i0, i1 = _circuit_args(layers)
    size = gkr_mle_size(layer_index)
    add, mul = np.empty((size, 4), dtype=np.uint64), np.empty((size, 4), dtype=np.uint64)
    assert lib().ora_circuit_add_mult_mle(nl, ng, _p(gt), _p(i0), _p(i1), C.c_size_t(layer_index), _p(add), _p(mul)) == 0
    return add, mul


def gkr_prove(layers, evaluation):
    nl, ng, gt, i0, i1 = _circuit_args(layers)
    flat = np.ascontiguousarray(np.concatenate([_fr(e).reshape(-1, 4) for e in evaluation]))
    lens = (C.c_size_t * len(evaluation))(*[len(e) for e in evaluation])
    proof = GkrProof()
    rc = lib().ora_gkr_prove(nl, ng, _p(gt), _p(i0), _p(i1), _p(flat), lens, C.byref(proof))
    if rc != 0:
        raise AssertionError("gkr_prove: shape error %d" % rc)
    return proof


def gkr_prove_sparse(layers, evaluation):
    """GKRProtocol::prove on sparse containers (gkr_sparse.c): the reference's (b, c)-table prover in O(gates) per round"""
    nl, ng, gt, i0, i1 = _circuit_args(layers)
    flat = np.ascontiguousarray(np.concatenate([_fr(e).reshape(-1, 4) for e in evaluation]))
    lens = (C.c_size_t * len(evaluation))(*[len(e) for e in evaluation])
    proof = GkrProof()
    rc = lib().ora_gkr_prove_sparse(nl, ng, _p(gt), _p(i0), _p(i1), _p(flat), lens, C.byref(proof))
    if rc != 0:
        raise AssertionError("gkr_prove_sparse: shape error %d" % rc)
    return proof


def gkr_verify(layers, inp, proof):
    inp = _fr(inp).reshape(-1, 4)
    nl, ng, gt, i0, i1 = _circuit_args(layers)
    return lib().ora_gkr_verify(nl, ng, _p(gt), _p(i0), _p(i1), _p(inp), C.c_size_t(inp.shape[0]), C.byref(proof)) == 1


# ---- NTT ---------------------------------------------------------------------------------------
def domain_fft(coeffs, size):
    coeffs = _fr(coeffs).reshape(-1, 4)
    o = np.empty((size, 4), dtype=np.uint64)
    assert lib().ora_domain_fft(_p(o), _p(coeffs), C.c_size_t(coeffs.shape[0]), C.c_size_t(size)) == 0
    return o


def domain_ifft(evals, size):
    evals = _fr(evals).reshape(-1, 4)
    o = np.empty((size, 4), dtype=np.uint64)
    assert lib().ora_domain_ifft(_p(o), _p(evals), C.c_size_t(evals.shape[0]), C.c_size_t(size)) == 0
    return o


def univariate_multiply(a, b):
    a, b = _fr(a).reshape(-1, 4), _fr(b).reshape(-1, 4)
    o = np.empty((a.shape[0] + b.shape[0] - 1, 4), dtype=np.uint64)
    assert lib().ora_univariate_multiply(_p(o), _p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0])) == 0
    return o


def dense_mul(a, b):
    a, b = _fr(a).reshape(-1, 4), _fr(b).reshape(-1, 4)
    o = np.empty((a.shape[0] + b.shape[0], 4), dtype=np.uint64)
    n = lib().ora_dense_mul(_p(o), _p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0]))
    return o[:n].copy()


def dense_evaluate(coeffs, x):
    coeffs, x = _fr(coeffs).reshape(-1, 4), _fr(x).reshape(4)
    o = np.empty(4, dtype=np.uint64)
    lib().ora_dense_evaluate(_p(o), _p(coeffs), C.c_size_t(coeffs.shape[0]), _p(x))
    return o

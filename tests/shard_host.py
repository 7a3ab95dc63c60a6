"""Test harness: the library's exchange protocols (csrc/shard_protocol.hpp) compiled for the HOST (tests/cpp/shard_protocol_host.cpp)
and driven with Python checker engines + a torch.distributed (gloo) all-gather -- the same C++ protocol code libzkhip runs on its HIP
engines, exercised with world_size > 1 on a box without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "shard_protocol_host.cpp")
HDR = os.path.join(ROOT, "zk-cryptography_amd", "csrc", "shard_protocol.hpp")

U64P = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def build(out_dir):
    """g++ -shared build of the host protocol library; returns its path (call once, before spawning ranks)."""
    so = os.path.join(str(out_dir), "libshard_protocol_host.so")
    if not os.path.exists(so) or max(os.path.getmtime(SRC), os.path.getmtime(HDR)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so, SRC])
    return so


def _arr(ptr, *shape):
    """int64 torch view of host memory the protocol owns (field elements as 4 x uint64)"""
    n = int(np.prod(shape))
    a = np.ctypeslib.as_array(C.cast(ptr, U64P), shape=(n,))
    return torch.from_numpy(a.view(np.int64)).view(*shape)


def _claimed(ptr):
    return None if not ptr else np.ctypeslib.as_array(C.cast(ptr, U64P), shape=(4,)).copy()


class _Guard:
    """callbacks must not raise through C frames: remember the exception, return a status"""

    def __init__(self):
        self.error = None

    def wrap(self, restype, fn, fail):
        def inner(*a):
            try:
                r = fn(*a)
                return 0 if r is None else r
            except BaseException as e:     # noqa: BLE001
                if self.error is None:
                    self.error = e
                return fail
        return inner


def _all_gather_cb(dist, world, guard):
    def ag(send, recv, nbytes):
        s = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)))
        r = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,)))
        dist.all_gather_into_tensor(r, s.clone())
    return C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)(guard.wrap(C.c_int, ag, 1))


SC_FIELDS = [("local_len", C.CFUNCTYPE(C.c_size_t)), ("use_stages", C.CFUNCTYPE(C.c_int)), ("tail_capacity", C.CFUNCTYPE(C.c_uint32)),
             ("overlap_plan", C.CFUNCTYPE(C.c_int, C.c_uint32, C.c_size_t, u32p, u32p, u32p, C.c_int)),
             ("overlap_sums", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)),
             ("overlap_rounds1", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32)),
             ("overlap_rounds2", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)),
             ("stage_plan", C.CFUNCTYPE(C.c_int, C.c_uint32, C.c_size_t, u32p, C.c_int)),
             ("stage_block_sums", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)),
             ("stage_absorb", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t)),
             ("stage_fold", C.CFUNCTYPE(C.c_int)),
             ("local_half_sums", C.CFUNCTYPE(C.c_int, C.c_void_p)),
             ("absorb", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p)),
             ("fold", C.CFUNCTYPE(C.c_int)),
             ("local_table", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)),
             ("tail", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p)),
             ("all_gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t))]


class ScCallbacks(C.Structure):
    _fields_ = SC_FIELDS


MC_FIELDS = [("local_len", C.CFUNCTYPE(C.c_size_t)), ("tail_capacity", C.CFUNCTYPE(C.c_uint32)), ("record_len", C.CFUNCTYPE(C.c_uint32)),
             ("table_count", C.CFUNCTYPE(C.c_uint32)),
             ("stage_record_len", C.CFUNCTYPE(C.c_int, C.c_uint32, C.c_size_t, u32p, C.c_int)),
             ("stage_sums", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32)),
             ("stage_absorb", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)),
             ("round_sums", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32)),
             ("absorb", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)),
             ("local_tables", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_size_t)),
             ("tail", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)),
             ("all_gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t))]


class McCallbacks(C.Structure):
    _fields_ = MC_FIELDS


ERR_PEER = -7


class RankFailed(Exception):
    """the protocol returned a status: .rc (ERR_PEER on the healthy ranks, the injected status on the failed one), .exchanges"""

    def __init__(self, rc, exchanges):
        super().__init__("protocol status %d after %d exchanges" % (rc, exchanges))
        self.rc, self.exchanges = rc, exchanges


def prove_sumcheck(so, e, world, dist, claimed_sum=None, inject=None):
    """zkshard::sumcheck_prove over checker engine `e` (the interface of tests/test_distributed_cpu.OracleSumcheckEngine);
    returns (engine.finish(total rounds), exchanges).  inject = (exchange index, status): this rank fails there (raises RankFailed,
    as every other rank does with ERR_PEER).  The plan callbacks take the protocol's n_local: they are called after a failure too."""
    lib = C.CDLL(so)
    g = _Guard()
    total_rounds = (e.local_len() * world).bit_length() - 1
    st = {}

    def overlap_plan(w, n_local, k1, k2, mid, failed):
        plan = e.overlap_plan(w, n_local) if hasattr(e, "overlap_plan") else None
        k1[0], k2[0], mid[0] = plan if plan else (0, 0, 0)
        st["k1"], st["k2"] = k1[0], k2[0]

    def stage_plan(w, n_local, k, failed):
        k[0] = e.stage_plan(w, n_local) if hasattr(e, "stage_plan") else 0

    fns = {
        "local_len": (lambda: e.local_len(), 0),
        "use_stages": (lambda: 1 if getattr(e, "use_stages", True) else 0, 0),
        "tail_capacity": (lambda: e.tail_capacity(), 0),
        "overlap_plan": (overlap_plan, 1),
        "overlap_sums": (lambda out, n: e.overlap_sums(_arr(out, n, 4)), 1),
        "overlap_rounds1": (lambda gth, w, cl, mid_out, mid: e.overlap_rounds1(_arr(gth, w, 1 << st["k1"], 4), w, _arr(mid_out, mid, 4), _claimed(cl)), 1),
        "overlap_rounds2": (lambda gth, w, mid: e.overlap_rounds2(_arr(gth, w, mid, 4), w), 1),
        "stage_plan": (stage_plan, 1),
        "stage_block_sums": (lambda out, n: e.stage_block_sums(_arr(out, n, 4)), 1),
        "stage_absorb": (lambda gth, w, cl, n: e.stage_absorb(_arr(gth, w, n, 4), w, _claimed(cl)), 1),
        "stage_fold": (lambda: e.stage_fold(), 1),
        "local_half_sums": (lambda out: e.local_half_sums(_arr(out, 2, 4)), 1),
        "absorb": (lambda gth, w, cl: e.absorb(_arr(gth, w, 2, 4), w, _claimed(cl)), 1),
        "fold": (lambda: e.fold(), 1),
        "local_table": (lambda out, n: e.local_table(_arr(out, n, 4)), 1),
        "tail": (lambda vals, m, cl: e.tail(_arr(vals, m, 4), m, _claimed(cl)), 1),
    }
    cb = ScCallbacks()
    keep = []
    for name, ftype in SC_FIELDS:
        if name == "all_gather":
            f = _all_gather_cb(dist, world, g)
        else:
            fn, fail = fns[name]
            f = ftype(g.wrap(None, fn, fail))
        keep.append(f)
        setattr(cb, name, f)
    ex = C.c_uint32(0)
    cs = np.ascontiguousarray(claimed_sum, dtype=np.uint64) if claimed_sum is not None else None
    inj_at, inj_rc = inject if inject is not None else (-1, 0)
    rc = lib.zkshard_host_sumcheck(C.byref(cb), C.c_uint32(world), cs.ctypes.data_as(C.c_void_p) if cs is not None else None, C.byref(ex),
                                   C.c_int(inj_at), C.c_int(inj_rc))
    if g.error is not None:
        raise g.error
    if rc != 0:
        raise RankFailed(rc, ex.value)
    return e.finish(total_rounds), ex.value


def prove_composed(so, e, world, dist, use_stages=None, inject=None):
    """zkshard::composed_prove over checker engine `e` (the interface of OracleComposedEngine); use_stages None = world > 1.
    inject as prove_sumcheck."""
    lib = C.CDLL(so)
    g = _Guard()
    total_rounds = (e.local_len() * world).bit_length() - 1
    nt = e.table_count()

    def stage_record_len(w, n_local, vals, failed):
        vals[0] = e.stage_record_len(n_local) if hasattr(e, "stage_record_len") else 0

    fns = {
        "local_len": (lambda: e.local_len(), 0),
        "tail_capacity": (lambda: e.tail_capacity(), 0),
        "record_len": (lambda: e.record_len(), 0),
        "table_count": (lambda: e.table_count(), 0),
        "stage_record_len": (stage_record_len, 1),
        "stage_sums": (lambda out, vals: e.stage_sums(_arr(out, vals, 4)), 1),
        "stage_absorb": (lambda gth, w, vals: e.stage_absorb(_arr(gth, w, vals, 4), w), 1),
        "round_sums": (lambda out, rec: e.round_sums(_arr(out, rec, 4)), 1),
        "absorb": (lambda gth, w, rec: e.absorb(_arr(gth, w, rec, 4), w), 1),
        "local_tables": (lambda out, n_t, n_local: e.local_tables(_arr(out, n_t, n_local, 4)), 1),
        "tail": (lambda tabs, m, n_t: e.tail(_arr(tabs, n_t, m, 4), m), 1),
    }
    cb = McCallbacks()
    keep = []
    for name, ftype in MC_FIELDS:
        if name == "all_gather":
            f = _all_gather_cb(dist, world, g)
        else:
            fn, fail = fns[name]
            f = ftype(g.wrap(None, fn, fail))
        keep.append(f)
        setattr(cb, name, f)
    ex = C.c_uint32(0)
    stages = (world > 1) if use_stages is None else bool(use_stages)
    inj_at, inj_rc = inject if inject is not None else (-1, 0)
    rc = lib.zkshard_host_composed(C.byref(cb), C.c_uint32(world), C.c_int(1 if stages else 0), C.byref(ex), C.c_int(inj_at), C.c_int(inj_rc))
    if g.error is not None:
        raise g.error
    if rc != 0:
        raise RankFailed(rc, ex.value)
    return e.finish(total_rounds), ex.value

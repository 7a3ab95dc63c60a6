"""Multi-GPU host side: one process per GPU, every exchange an all-gather on the context's stream (RCCL over xGMI).

The exchange PROTOCOLS of the sharded provers -- stage plans, exchange order, tail gather, the per-layer GKR loop, the commit merge --
live inside libzkhip (csrc/shard_protocol.hpp, csrc/shard.hip; include/zkhip.h "the sharded provers"); this module only binds them:

* Comm -- zkhip_comm: this rank's end of the exchange.  Transports: an RCCL communicator the library creates itself (when the
  torch.distributed group runs on the "nccl" backend: the 128-byte unique id travels through the group once), or a callback that
  stages the (tiny) payloads through host memory and any torch.distributed backend (gloo in the tests), or none (world == 1).
* ShardedSumcheck -- zkhip_sc_prove_sharded: Sumcheck::prove over a table partitioned by the LOW index bits (rank g holds entry
  j*world + g at local index j; rounds fold the most significant variable, so every fold is local).  Shards of 2^19..2^24 entries
  take the overlapped stage (2^24 per rank on 8 ranks: 6 | 10 | 11 rounds and three exchanges), others the stage / round forms.
* ShardedComposedSumcheck -- zkhip_mc_prove_sharded: ComposedSumcheck::prove / MultiComposedSumcheckProver::prove_partial over
  shards, two rounds per exchange where every term is a product of two tables.
* sharded_commit -- zkhip_kzg_commit_sharded: (scalars, SRS points) split the same way, a sub-MSM per rank, one all-gather of the
  partial commitments, their group sum on every rank.
* HipSumcheckEngine / HipComposedEngine -- the split-phase sessions (zkhip_sc_* / zkhip_mc_*) the protocols run on; the tests also
  drive several of them in lockstep on one GPU.
"""
import ctypes as C

import numpy as np

from zk_cryptography_amd import _native as N

ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Comm:
    """zkhip_comm (include/zkhip.h): one per (context, group); cached, closed explicitly (Comm.close_all) or never."""

    _cache = []
    _pick_rccl_anyway = False     # tests: pick the library's RCCL communicator whatever backend torch.distributed runs (the fallback path)

    def __init__(self, ctx, world=1, rank=0, dist=None, group=None, transport=None):
        """transport: None = pick ("none" for world 1, "rccl" when `dist` runs the nccl backend, else "staged"); or a callable
        fn(user, d_send, d_recv, bytes_per_rank, stream) -> 0: the caller's own all-gather (zkhip_all_gather_fn)."""
        if world & (world - 1):
            raise AssertionError("world size must be a power of two (the tables have 2^n entries)")
        self.ctx, self.world, self.rank, self.dist, self.group = ctx, world, rank, dist, group
        self.handle = C.c_void_p()
        self.error = None
        self.fallback_reason = None          # set when the library's RCCL communicator was picked but did not come up on every rank
        auto = transport is None
        if auto:
            transport = "none" if world == 1 else ("rccl" if (_is_nccl(dist, group) or Comm._pick_rccl_anyway) else "staged")
        self.transport = transport
        lib = N.lib()
        if callable(transport):
            self._cb = transport if isinstance(transport, ALL_GATHER_FN) else ALL_GATHER_FN(transport)
            N.check(lib.zkhip_comm_create(ctx.handle, C.c_uint32(rank), C.c_uint32(world), self._cb, None, C.byref(self.handle)), "comm_create")
        elif transport == "none":
            assert world == 1
            N.check(lib.zkhip_comm_create(ctx.handle, C.c_uint32(0), C.c_uint32(1), None, None, C.byref(self.handle)), "comm_create")
        elif transport == "rccl":
            picked = auto and world > 1
            err = self._create_rccl(lib, raise_errors=not picked)
            if picked:
                # The library's own communicator at world > 1.  The ranks AGREE (over torch.distributed) twice before anything depends
                # on it: first that every rank CREATED its communicator -- a rank that did not must not leave its peers alone in the
                # probe's all-gather --, then that one small all-gather through it delivered every rank's bytes to every rank.  If
                # either answer is no on any rank, all of them fall back to the staged transport.
                created = self._all_ranks(err is None)
                ok = created and self._all_ranks(self._probe())
                if not ok:
                    self.close()
                    self.handle = C.c_void_p()
                    self.transport = "staged"
                    self.fallback_reason = (repr(err) if err is not None else
                                            ("the library's RCCL communicator could not be created on another rank" if not created else
                                             "probe all-gather through the library's RCCL communicator failed on some rank"))
                    self._create_staged(lib)
        else:
            self._create_staged(lib)
        if self.handle:
            ctx._comms.append(self)

    def _create_staged(self, lib):
        self._cb = ALL_GATHER_FN(self._staged_all_gather)      # kept alive with the comm
        N.check(lib.zkhip_comm_create(self.ctx.handle, C.c_uint32(self.rank), C.c_uint32(self.world), self._cb, None, C.byref(self.handle)), "comm_create")

    def _create_rccl(self, lib, raise_errors):
        """ncclGetUniqueId on rank 0, its broadcast, ncclCommInitRank on every rank.  -> None or the exception (raise_errors=False)"""
        try:
            uid = np.zeros(128, dtype=np.uint8)
            first = None
            if self.rank == 0:
                try:
                    N.check(lib.zkhip_rccl_unique_id(uid.ctypes.data_as(C.c_void_p)), "rccl_unique_id (librccl.so not found?)")
                except Exception as e:          # the broadcast below must still happen: the other ranks wait in it
                    first = e
            if self.world > 1:
                box = [uid.tobytes() if first is None else None]
                self.dist.broadcast_object_list(box, src=_global_rank(self.dist, self.group, 0), group=self.group)
                if box[0] is None:
                    raise first or RuntimeError("rank 0 could not create an RCCL unique id")
                uid = np.frombuffer(box[0], dtype=np.uint8).copy()
            elif first is not None:
                raise first
            N.check(lib.zkhip_comm_create_rccl(self.ctx.handle, uid.ctypes.data_as(C.c_void_p), C.c_uint32(self.rank), C.c_uint32(self.world),
                                               C.byref(self.handle)), "comm_create_rccl")
            return None
        except Exception as e:
            if raise_errors:
                raise
            return e

    def _probe(self):
        """one 64-byte all-gather through the communicator: rank r sends 64 bytes of value r"""
        try:
            import torch
            dev = getattr(self.ctx, "device", None) or torch.device("cuda")
            send = torch.full((64,), self.rank, dtype=torch.uint8, device=dev)
            recv = torch.zeros(64 * self.world, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            st = N.lib().zkhip_comm_all_gather(self.handle, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), C.c_size_t(64))
            if st != N.ZKHIP_OK:
                return False
            got = np.empty(64 * self.world, dtype=np.uint8)
            N.check(N.lib().zkhip_memcpy_d2h(self.ctx.handle, got.ctypes.data_as(C.c_void_p), C.c_void_p(recv.data_ptr()), C.c_size_t(64 * self.world)), "d2h")
            return bool(np.array_equal(got, np.repeat(np.arange(self.world, dtype=np.uint8), 64)))
        except Exception:
            return False

    def _all_ranks(self, ok):
        """True iff `ok` on every rank of the group"""
        import torch
        dev = "cuda" if _is_nccl(self.dist, self.group) else "cpu"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()) == 1)

    def _staged_all_gather(self, user, d_send, d_recv, nbytes, stream):
        """all-gather through host memory and torch.distributed (payloads are at most a few hundred KiB): waits for the stream.
        The collective itself runs on tensors the group's backend accepts: an nccl group has no CPU backend, so there the staged
        bytes go up into CUDA tensors of torch's own, through torch's all-gather, and come down again (this is the transport of
        last resort of a job whose library-side RCCL communicator did not come up -- correct first, 3 copies more)."""
        try:
            import torch
            lib = N.lib()
            send = torch.empty(nbytes, dtype=torch.uint8)
            N.check(lib.zkhip_memcpy_d2h(self.ctx.handle, C.c_void_p(send.data_ptr()), C.c_void_p(d_send), C.c_size_t(nbytes)), "d2h")
            if _is_nccl(self.dist, self.group):
                dev = getattr(self.ctx, "device", None) or torch.device("cuda")
                recv_d = torch.empty(nbytes * self.world, dtype=torch.uint8, device=dev)
                self.dist.all_gather_into_tensor(recv_d, send.to(dev), group=self.group)
                recv = recv_d.cpu()                                  # waits for the collective
            else:
                recv = torch.empty(nbytes * self.world, dtype=torch.uint8)
                self.dist.all_gather_into_tensor(recv, send, group=self.group)
            N.check(lib.zkhip_memcpy_h2d(self.ctx.handle, C.c_void_p(d_recv), C.c_void_p(recv.data_ptr()), C.c_size_t(nbytes * self.world)), "h2d")
            return 0
        except BaseException as e:      # never unwind through the C frames
            self.error = e
            return 1

    @classmethod
    def get(cls, ctx, world=1, rank=None, dist=None, group=None):
        if rank is None:
            rank = dist.get_rank(group) if (world > 1 and dist is not None) else 0
        # the cache holds the objects themselves and compares with `is`: an id() can be reused by a later context or group
        for m in cls._cache:
            if m.handle and m.ctx is ctx and m.world == world and m.rank == rank and (world == 1 or (m.dist is dist and m.group is group)):
                return m
        m = Comm(ctx, world, rank, dist, group)
        cls._cache.append(m)
        return m

    def stats(self):
        ex, by = C.c_uint64(0), C.c_uint64(0)
        N.check(N.lib().zkhip_comm_stats(self.handle, C.byref(ex), C.byref(by)), "comm_stats")
        return ex.value, by.value

    def measure(self, nbytes, iters=200):
        """(microseconds per exchange back to back, with a host wait per exchange), measured inside the library"""
        a, b = C.c_double(0), C.c_double(0)
        N.check(N.lib().zkhip_comm_measure(self.handle, C.c_size_t(nbytes), C.c_uint32(iters), C.byref(a), C.byref(b)), "comm_measure")
        return a.value, b.value

    def inject_failure(self, exchange_index, status=N.ERR_NOMEM):
        """test hook (zkhip_comm_inject_failure): the next protocol run on this comm fails on THIS rank in front of that exchange"""
        N.check(N.lib().zkhip_comm_inject_failure(self.handle, C.c_int(exchange_index), C.c_int(status)), "comm_inject_failure")

    def check(self, status, what):
        """N.check, re-raising what a staged exchange caught inside its callback"""
        if status != N.ZKHIP_OK and self.error is not None:
            e, self.error = self.error, None
            raise e
        N.check(status, what)

    def close(self):
        if self.handle:
            h, self.handle = self.handle, None
            N.lib().zkhip_comm_destroy(h)
        if self in self.ctx._comms:
            self.ctx._comms.remove(self)
        if self in Comm._cache:
            Comm._cache.remove(self)

    @classmethod
    def close_all(cls):
        for m in list(cls._cache):
            m.close()
        del cls._cache[:]


def _is_nccl(dist, group):
    try:
        return dist is not None and hasattr(dist, "get_backend") and str(dist.get_backend(group)) == "nccl"
    except Exception:
        return False


def _global_rank(dist, group, group_rank):
    return dist.get_global_rank(group, group_rank) if group is not None else group_rank


def shard_interleaved(full, rank, world):
    """rank's shard of a full table / SRS: entries rank, rank + world, ... (works on numpy and torch)."""
    return full[rank::world]


class HipSumcheckEngine:
    """Split-phase prover state on this rank's GPU (zkhip_sc_* in include/zkhip.h)."""

    def __init__(self, local_table):
        import torch
        self.torch = torch
        self.table = local_table          # int64 [n_local, 4] CUDA tensor, kept alive for the duration
        self.ctx = N.Context.get(local_table.device.index)
        self.st = C.c_void_p()
        N.check(N.lib().zkhip_sc_begin(self.ctx.handle, N.ptr(local_table), C.c_size_t(local_table.shape[0]),
                                       C.byref(self.st)), "sc_begin")

    def abort(self):
        """Releases the device state without a result (zkhip_sc_abort): error paths and dropped engines."""
        if getattr(self, "st", None):
            st, self.st = self.st, None
            try:
                N.lib().zkhip_sc_abort(st)
            except Exception:       # interpreter shutdown: the library may already be gone
                pass

    __del__ = abort

    def new_buffer(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.int64, device=self.table.device)

    def local_len(self):
        n = C.c_size_t(0)
        N.check(N.lib().zkhip_sc_local_len(self.st, C.byref(n)), "sc_local_len")
        return n.value

    def local_half_sums(self, out):
        N.check(N.lib().zkhip_sc_local_half_sums(self.st, N.ptr(out)), "sc_local_half_sums")

    def absorb(self, gathered, world, claimed_sum=None):
        cs = np.ascontiguousarray(claimed_sum, dtype=np.uint64) if claimed_sum is not None else None
        N.check(N.lib().zkhip_sc_absorb(self.st, N.ptr(gathered), C.c_uint32(world),
                                        cs.ctypes.data_as(C.c_void_p) if cs is not None else None), "sc_absorb")

    def fold(self):
        N.check(N.lib().zkhip_sc_fold(self.st), "sc_fold")

    def local_value(self, out):
        N.check(N.lib().zkhip_sc_local_value(self.st, N.ptr(out)), "sc_local_value")

    def stage_plan(self, world):
        k = C.c_uint32(0)
        N.check(N.lib().zkhip_sc_stage_plan(self.st, C.c_uint32(world), C.byref(k)), "sc_stage_plan")
        return k.value

    def stage_block_sums(self, out):
        N.check(N.lib().zkhip_sc_stage_block_sums(self.st, N.ptr(out)), "sc_stage_block_sums")

    def stage_absorb(self, gathered, world, claimed_sum=None):
        cs = np.ascontiguousarray(claimed_sum, dtype=np.uint64) if claimed_sum is not None else None
        N.check(N.lib().zkhip_sc_stage_absorb(self.st, N.ptr(gathered), C.c_uint32(world),
                                              cs.ctypes.data_as(C.c_void_p) if cs is not None else None), "sc_stage_absorb")

    def stage_fold(self):
        N.check(N.lib().zkhip_sc_stage_fold(self.st), "sc_stage_fold")

    def overlap_plan(self, world):
        """(k1, k2, mid_entries) of the overlapped stage, or None when it does not apply to this shard"""
        k1, k2, mid = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        N.check(N.lib().zkhip_sc_overlap_plan(self.st, C.c_uint32(world), C.byref(k1), C.byref(k2), C.byref(mid)), "sc_overlap_plan")
        return (k1.value, k2.value, mid.value) if k1.value else None

    def overlap_sums(self, out):
        N.check(N.lib().zkhip_sc_overlap_sums(self.st, N.ptr(out)), "sc_overlap_sums")

    def overlap_rounds1(self, gathered, world, mid_out, claimed_sum=None):
        cs = np.ascontiguousarray(claimed_sum, dtype=np.uint64) if claimed_sum is not None else None
        N.check(N.lib().zkhip_sc_overlap_rounds1(self.st, N.ptr(gathered), C.c_uint32(world),
                                                 cs.ctypes.data_as(C.c_void_p) if cs is not None else None, N.ptr(mid_out)), "sc_overlap_rounds1")

    def overlap_rounds2(self, gathered, world):
        N.check(N.lib().zkhip_sc_overlap_rounds2(self.st, N.ptr(gathered), C.c_uint32(world)), "sc_overlap_rounds2")

    def local_table(self, out):
        N.check(N.lib().zkhip_sc_local_table(self.st, N.ptr(out)), "sc_local_table")

    def tail_capacity(self):
        return N.lib().zkhip_sc_tail_capacity()

    def tail(self, values, m, claimed_sum=None):
        cs = np.ascontiguousarray(claimed_sum, dtype=np.uint64) if claimed_sum is not None else None
        N.check(N.lib().zkhip_sc_tail(self.st, N.ptr(values), C.c_uint32(m),
                                      cs.ctypes.data_as(C.c_void_p) if cs is not None else None), "sc_tail")

    def finish(self, n_rounds):
        s = np.empty(4, dtype=np.uint64)
        rp = np.empty((max(n_rounds, 1), 2, 4), dtype=np.uint64)
        ch = np.empty((max(n_rounds, 1), 4), dtype=np.uint64)
        got = C.c_uint32(0)
        st, self.st = self.st, None            # zkhip_sc_finish releases the state whatever it returns
        N.check(N.lib().zkhip_sc_finish(st, s.ctypes.data_as(C.c_void_p), rp.ctypes.data_as(C.c_void_p),
                                        ch.ctypes.data_as(C.c_void_p), C.byref(got)), "sc_finish")
        assert got.value == n_rounds
        return s, rp[:n_rounds], ch[:n_rounds]


class ShardedSumcheck:
    """Sumcheck::prove (sumcheck/src/sumcheck.rs:29-61) over a table sharded by low index bits: zkhip_sc_prove_sharded on the
    engine's session.  Every rank returns the same (sum, round_polys [n_vars, 2, 4], challenges [n_vars, 4]) -- the values a
    single-GPU / reference prover yields on the full table; `exchanges` = all-gathers issued."""

    def __init__(self, engine, world=1, group=None, dist=None, comm=None):
        if world & (world - 1):
            raise AssertionError("world size must be a power of two (the table has 2^n entries)")
        self.e = engine
        self.world = world
        self.comm = comm or Comm.get(engine.ctx, world, None, dist, group)
        self.exchanges = 0

    def prove(self, claimed_sum=None):
        e = self.e
        n_rounds = (e.local_len() * self.world).bit_length() - 1
        s = np.empty(4, dtype=np.uint64)
        rp = np.empty((max(n_rounds, 1), 2, 4), dtype=np.uint64)
        ch = np.empty((max(n_rounds, 1), 4), dtype=np.uint64)
        cs = np.ascontiguousarray(claimed_sum, dtype=np.uint64) if claimed_sum is not None else None
        ex = C.c_uint32(0)
        st, e.st = e.st, None              # the call finishes (releases) the session whatever it returns
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        self.comm.check(N.lib().zkhip_sc_prove_sharded(st, self.comm.handle, p(cs) if cs is not None else None, p(s), p(rp), p(ch),
                                                       C.byref(ex)), "sc_prove_sharded")
        self.exchanges = ex.value
        return s, rp[:n_rounds], ch[:n_rounds]


class HipComposedEngine:
    """Split-phase composed / multi-composed prover state on this rank's GPU (zkhip_mc_* in include/zkhip.h).

    terms: list of lists of int64 [n_local, 4] CUDA tensors (this rank's shard of every table, term after term);
    multi = False: ComposedSumcheck (one term);  multi = True: MultiComposedSumcheckProver::prove_partial with the
    claimed sum of the whole tables."""

    def __init__(self, terms, world, multi, claimed_sum=None, ctx=None, lin=None, cont=False, out_base=0):
        """lin: per term an additive table (or None) -- term = product of its tables + that table; cont / out_base: continue
        the previous session's transcript and recorded rounds from round out_base on (zkhip_mc_begin_ex)."""
        import torch
        self.torch = torch
        self.tables = [t for term in terms for t in term]      # kept alive for the duration
        self.lin = list(lin) if lin is not None else None
        self.sizes = [len(term) for term in terms]
        self.multi = bool(multi)
        self.device = self.tables[0].device
        self.ctx = ctx or N.Context.get(self.device.index)     # one session per context at a time
        self.st = C.c_void_p()
        cs = np.ascontiguousarray(claimed_sum, dtype=np.uint64).reshape(4) if claimed_sum is not None else None
        self.sum = cs
        ptrs = (C.c_void_p * len(self.tables))(*[t.data_ptr() for t in self.tables])
        lin_ptrs = (C.c_void_p * len(self.sizes))(*[t.data_ptr() if t is not None else None for t in self.lin]) if self.lin else None
        N.check(N.lib().zkhip_mc_begin_ex(self.ctx.handle, ptrs, (C.c_uint32 * len(self.sizes))(*self.sizes),
                                          C.c_uint32(len(self.sizes)), lin_ptrs, C.c_size_t(self.tables[0].shape[0]), C.c_uint32(world),
                                          C.c_int(1 if multi else 0), cs.ctypes.data_as(C.c_void_p) if cs is not None else None,
                                          C.c_int(1 if cont else 0), C.c_uint32(out_base), C.byref(self.st)), "mc_begin")
        rec, nt = C.c_uint32(0), C.c_uint32(0)
        N.check(N.lib().zkhip_mc_record_len(self.st, C.byref(rec), C.byref(nt)), "mc_record_len")
        self.rec, self.n_tables = rec.value, nt.value

    def abort(self):
        """Releases the device state without a result (zkhip_mc_abort): error paths and dropped engines."""
        if getattr(self, "st", None):
            st, self.st = self.st, None
            try:
                N.lib().zkhip_mc_abort(st)
            except Exception:       # interpreter shutdown: the library may already be gone
                pass

    __del__ = abort

    def new_buffer(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.int64, device=self.device)

    def record_len(self):
        return self.rec

    def table_count(self):
        return self.n_tables

    def local_len(self):
        n = C.c_size_t(0)
        N.check(N.lib().zkhip_mc_local_len(self.st, C.byref(n)), "mc_local_len")
        return n.value

    def tail_capacity(self):
        cap = C.c_uint32(0)
        N.check(N.lib().zkhip_mc_tail_capacity(self.st, C.byref(cap)), "mc_tail_capacity")
        return cap.value

    def stage_record_len(self):
        """field elements of the two-round stage record (zkhip_mc_stage_*), 0 when the next step cannot be a stage"""
        v = C.c_uint32(0)
        N.check(N.lib().zkhip_mc_stage_record_len(self.st, C.byref(v)), "mc_stage_record_len")
        return v.value

    def stage_sums(self, out):
        N.check(N.lib().zkhip_mc_stage_sums(self.st, N.ptr(out)), "mc_stage_sums")

    def stage_absorb(self, gathered, world):
        N.check(N.lib().zkhip_mc_stage_absorb(self.st, N.ptr(gathered), C.c_uint32(world)), "mc_stage_absorb")

    def round_sums(self, out):
        N.check(N.lib().zkhip_mc_round_sums(self.st, N.ptr(out)), "mc_round_sums")

    def absorb(self, gathered, world):
        N.check(N.lib().zkhip_mc_absorb(self.st, N.ptr(gathered), C.c_uint32(world)), "mc_absorb")

    def local_tables(self, out):
        N.check(N.lib().zkhip_mc_local_tables(self.st, N.ptr(out)), "mc_local_tables")

    def tail(self, tables, m):
        N.check(N.lib().zkhip_mc_tail(self.st, N.ptr(tables), C.c_uint32(m)), "mc_tail")

    def finish(self, n_rounds):
        """-> (round polynomials, challenges [n_rounds, 4]); round polynomials are uint64 [n_rounds, K+1, 4] evaluations
        (ComposedSumcheck) or a list of (coeffs [m, 4], pows [m, 4]) pairs (multi-composed)."""
        ch = np.empty((n_rounds, 4), dtype=np.uint64)
        if not self.multi:
            rp = np.empty((n_rounds, self.sizes[0] + 1, 4), dtype=np.uint64)
            lens = None
        else:
            rp = np.zeros((n_rounds, 7, 2, 4), dtype=np.uint64)
            lens = np.zeros(n_rounds, dtype=np.uint32)
        st, self.st = self.st, None            # zkhip_mc_finish releases the state whatever it returns
        N.check(N.lib().zkhip_mc_finish(st, lens.ctypes.data_as(C.c_void_p) if lens is not None else None,
                                        rp.ctypes.data_as(C.c_void_p), ch.ctypes.data_as(C.c_void_p)), "mc_finish")
        if not self.multi:
            return rp, ch
        return [(rp[r, : lens[r], 0].copy(), rp[r, : lens[r], 1].copy()) for r in range(n_rounds)], ch


class ShardedComposedSumcheck:
    """ComposedSumcheck::prove (composed_sumcheck.rs:32-67) / MultiComposedSumcheckProver::prove_partial
    (multi_composed_sumcheck.rs:56-121) over tables sharded by low index bits ("GKR tables"): zkhip_mc_prove_sharded on
    the engine's session.  One exchange per round, or ONE PER TWO ROUNDS for claims whose terms are products of two tables
    (use_stages; default: on when world > 1 -- stages save exchanges, not work).  Every rank returns what a single-GPU / reference
    prover produces on the whole tables."""

    def __init__(self, engine, world=1, group=None, dist=None, use_stages=None, comm=None):
        if world & (world - 1):
            raise AssertionError("world size must be a power of two (the tables have 2^n entries)")
        self.e = engine
        self.world = world
        self.comm = comm or Comm.get(engine.ctx, world, None, dist, group)
        self.use_stages = -1 if use_stages is None else int(bool(use_stages))
        self.exchanges = 0

    def prove(self, collect=True, finish_rounds=None):
        """collect=False: run the rounds and release the session without reading anything back (a later session that
        continues it delivers the rounds of both); finish_rounds: how many recorded rounds are read (default: this session's)."""
        e = self.e
        total_rounds = (e.local_len() * self.world).bit_length() - 1
        n_rounds = finish_rounds if finish_rounds is not None else total_rounds
        ex = C.c_uint32(0)
        st, e.st = e.st, None              # finished (released) by the call whatever it returns
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        if not collect:
            self.comm.check(N.lib().zkhip_mc_prove_sharded(st, self.comm.handle, C.c_int(self.use_stages), None, None, None, C.byref(ex)),
                            "mc_prove_sharded")
            self.exchanges = ex.value
            return None
        ch = np.empty((n_rounds, 4), dtype=np.uint64)
        if not e.multi:
            rp = np.empty((n_rounds, e.sizes[0] + 1, 4), dtype=np.uint64)
            lens = None
        else:
            rp = np.zeros((n_rounds, 7, 2, 4), dtype=np.uint64)
            lens = np.zeros(n_rounds, dtype=np.uint32)
        self.comm.check(N.lib().zkhip_mc_prove_sharded(st, self.comm.handle, C.c_int(self.use_stages), p(lens) if lens is not None else None,
                                                       p(rp), p(ch), C.byref(ex)), "mc_prove_sharded")
        self.exchanges = ex.value
        if not e.multi:
            return rp, ch
        return [(rp[r, : lens[r], 0].copy(), rp[r, : lens[r], 1].copy()) for r in range(n_rounds)], ch


def sharded_commit(points_xy, inf, scalars, comm, table=None, require_equal_len=True):
    """KZG commit over (scalars, SRS) sharded across the ranks of `comm` (zkhip_kzg_commit_sharded): this rank's shard of the
    SRS (points_xy int64 [n, 12] + inf uint8 [n], or its shifted-SRS `table`) and of the scalars (int64 [m, 4]) -> the full
    commitment (xy uint64[12], inf bool) on every rank."""
    xy = np.empty(12, dtype=np.uint64)
    oinf = C.c_uint8(0)
    st = N.lib().zkhip_kzg_commit_sharded(comm.handle, None if table is not None else N.ptr(points_xy), N.ptr(table) if table is not None else None,
                                          N.ptr(inf), C.c_size_t(inf.shape[0]), N.ptr(scalars), C.c_size_t(scalars.shape[0]),
                                          C.c_int(1 if require_equal_len else 0), xy.ctypes.data_as(C.c_void_p), C.byref(oinf))
    comm.check(st, "kzg_commit_sharded")
    return xy, bool(oinf.value)


def hip_sum_affine(xy, inf):
    out = np.empty(12, dtype=np.uint64)
    oinf = C.c_uint8(0)
    N.check(N.lib().zkhip_g1_sum_affine(xy.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(xy.shape[0]), out.ctypes.data_as(C.c_void_p), C.byref(oinf)), "g1_sum")
    return out, bool(oinf.value)

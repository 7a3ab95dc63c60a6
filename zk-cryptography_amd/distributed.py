"""Multi-GPU host logic: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

Only the hot path's two sharding schemes live here:

* ShardedSumcheck -- the evaluation table of N = n_local * world entries is partitioned by the LOW index
  bits (rank g holds entry j*world + g at local index j).  Sumcheck rounds fold variable 0 = the most
  significant index bit, so every fold is local.  Stage form: per k rounds the ranks all-gather their 2^k partial
  block sums (8 KiB for k = 8; modular addition is not an RCCL reduction, so the payload is gathered and added
  locally), run the k rounds on the summed block sums with a replicated transcript, and fold their shard by k
  variables locally; once the remaining table fits the tail (2048 entries) it is gathered and the last rounds run
  replicated.  Shards of 2^19..2^24 entries take the OVERLAPPED stage first (the single-GPU plan in exchange form):
  k1 rounds on coarse sums, then k2 rounds on the fine sums folded by those k1 challenges while the shard's
  k1-variable fold runs on the engine's fold stream -- the second exchange and the serial rounds hide behind the one
  pass over the shard; 2^24 per rank on 8 ranks is 6 | 10 | 11 rounds and three exchanges.  Round form (kept as a
  fallback): one 64-byte exchange per round.
* sharded_commit -- (scalars, SRS points) are split the same way; each rank runs a full sub-MSM and the
  `world` partial commitments (104 bytes each) are all-gathered and summed.

The per-rank compute sits behind a small "engine" interface so that the exchange protocol can be exercised
on CPU (gloo) in tests with a checker engine; the product engine below is the HIP one.
"""
import ctypes as C

import numpy as np

from zk_cryptography_amd import _native as N


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


def _all_gather(dist, group, out, inp, world):
    if world == 1:
        out.view(-1)[:] = inp.view(-1)
    else:
        dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=group)   # flat: rank-major concatenation


class ShardedSumcheck:
    """Sumcheck::prove (sumcheck/src/sumcheck.rs:29-61) over a table sharded by low index bits.

    Every rank returns the same (sum, round_polys [n_vars, 2, 4], challenges [n_vars, 4]) -- the values a
    single-GPU / reference prover yields on the full table."""

    def __init__(self, engine, world=1, group=None, dist=None):
        self.e = engine
        self.world = world
        self.group = group
        self.dist = dist
        if world & (world - 1):
            raise AssertionError("world size must be a power of two (the table has 2^n entries)")

    def prove(self, claimed_sum=None):
        try:
            return self._prove(claimed_sum)
        except BaseException:
            if hasattr(self.e, "abort"):
                self.e.abort()              # a failed collective / assert must not leave the context's workspace lent
            raise

    def _prove(self, claimed_sum=None):
        e, world = self.e, self.world
        n_local = e.local_len()
        total_rounds = (n_local * world).bit_length() - 1
        absorbed = False
        self.exchanges = 0                 # collectives of this prove (what a multi-GPU run pays on top of the kernels)
        plan = e.overlap_plan(world) if getattr(e, "use_stages", True) and hasattr(e, "overlap_plan") else None
        if plan:
            # overlapped stage (shards of 2^19..2^24 entries): k1 rounds on coarse block sums, then k2 rounds on the fine sums
            # folded by those k1 challenges WHILE the shard's k1-variable fold runs on the engine's fold stream -- the second
            # exchange and the serial rounds hide behind the one pass over the shard
            k1, k2, mid = plan
            mine = e.new_buffer(1 << k1, 4)
            e.overlap_sums(mine)
            gathered = e.new_buffer(world, 1 << k1, 4)
            _all_gather(self.dist, self.group, gathered, mine, world)          # C1
            mine = e.new_buffer(mid, 4)
            e.overlap_rounds1(gathered, world, mine, claimed_sum)
            gathered = e.new_buffer(world, mid, 4)
            _all_gather(self.dist, self.group, gathered, mine, world)          # C2, beside the fold
            e.overlap_rounds2(gathered, world)
            self.exchanges += 2
            absorbed = True
            n_local >>= k1 + k2
        if getattr(e, "use_stages", True) and hasattr(e, "stage_plan"):
            # stage form: one exchange per k rounds (32 * 2^k bytes per rank), then one local k-variable fold
            while True:
                k = e.stage_plan(world)
                if k == 0:
                    break
                mine = e.new_buffer(1 << k, 4)
                e.stage_block_sums(mine)
                gathered = e.new_buffer(world, 1 << k, 4)
                _all_gather(self.dist, self.group, gathered, mine, world)      # C1: RCCL all-gather over xGMI
                self.exchanges += 1
                e.stage_absorb(gathered, world, None if absorbed else claimed_sum)
                absorbed = True
                e.stage_fold()
                n_local >>= k
        # round form: one 64-byte exchange per round -- the whole protocol for engines without stages, and the way down
        # to the tail size where a stage no longer fits (shards of a few entries on many ranks)
        cap = e.tail_capacity()
        if n_local * world > cap and n_local > 1:
            send = e.new_buffer(2, 4)
            recv = e.new_buffer(world, 2, 4)
            while n_local * world > cap and n_local > 1:
                e.local_half_sums(send)
                _all_gather(self.dist, self.group, recv, send, world)
                self.exchanges += 1
                e.absorb(recv, world, None if absorbed else claimed_sum)       # local modular add + transcript -> challenge
                absorbed = True
                e.fold()                                                        # local: partners share the low index bits
                n_local //= 2
        if n_local * world > 1:
            # the whole remaining table now fits one workgroup's LDS: gather it and finish replicated
            mine = e.new_buffer(n_local, 4)
            e.local_table(mine)
            gathered = e.new_buffer(world, n_local, 4)
            _all_gather(self.dist, self.group, gathered, mine, world)
            self.exchanges += 1
            full = gathered.transpose(0, 1).contiguous().view(n_local * world, 4)   # entry j*world + g <- rank g, local j
            e.tail(full, n_local * world, None if absorbed else claimed_sum)
        return e.finish(total_rounds)


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
    (multi_composed_sumcheck.rs:56-121) over tables sharded by low index bits ("GKR tables").

    One exchange per round: a record of (K_p + 1) partial sums per term.  Claims whose terms are products of TWO tables (the
    reference's ComposedSumcheck bench shape and every GKR layer claim) take TWO rounds per exchange instead: a product does not
    commute with block sums, but it is bilinear in them, so the next two round polynomials are functions of the 16 cross-block
    sums per term (csrc/composed_stage.hpp) -- the record is 20 field elements per term, the ranks' records are added, two
    transcript rounds run replicated and every rank folds its shards by both challenges.  Every rank returns what engine.finish
    yields -- the round polynomials and challenges a single-GPU / reference prover produces on the whole tables."""

    def __init__(self, engine, world=1, group=None, dist=None, use_stages=None):
        self.e = engine
        self.world = world
        self.group = group
        self.dist = dist
        # stages save exchanges, not work (on one rank two fused rounds are cheaper than a stage: DESIGN.md section 6)
        self.use_stages = (world > 1) if use_stages is None else bool(use_stages)
        if world & (world - 1):
            raise AssertionError("world size must be a power of two (the tables have 2^n entries)")

    def prove(self, collect=True, finish_rounds=None):
        """collect=False: run the rounds and release the session without reading anything back (a later session that
        continues it delivers the rounds of both); finish_rounds: how many recorded rounds finish() reads (default: this
        session's)."""
        try:
            return self._prove(collect, finish_rounds)
        except BaseException:
            if hasattr(self.e, "abort"):
                self.e.abort()
            raise

    def _prove(self, collect=True, finish_rounds=None):
        e, world = self.e, self.world
        n_local = e.local_len()
        total_rounds = (n_local * world).bit_length() - 1
        self.exchanges = 0
        cap = e.tail_capacity()
        rec = e.record_len()
        if self.use_stages and hasattr(e, "stage_record_len"):
            send = recv = None
            while n_local * world > cap and n_local >= 4:
                vals = e.stage_record_len()
                if not vals:
                    break
                if send is None:
                    send, recv = e.new_buffer(vals, 4), e.new_buffer(world, vals, 4)
                e.stage_sums(send)                                            # 16 cross-block sums (+ 4 block sums) per term
                _all_gather(self.dist, self.group, recv, send, world)         # ONE exchange for two rounds
                self.exchanges += 1
                e.stage_absorb(recv, world)                                   # two transcript rounds + the fold by both challenges
                n_local //= 4
        if n_local * world > cap and n_local > 1:
            send = e.new_buffer(rec, 4)
            recv = e.new_buffer(world, rec, 4)
            while n_local * world > cap and n_local > 1:
                e.round_sums(send)                                            # fold at the previous challenge + partial sums
                _all_gather(self.dist, self.group, recv, send, world)         # RCCL all-gather over xGMI, <= 768 B per rank
                self.exchanges += 1
                e.absorb(recv, world)                                         # local modular add + transcript -> challenge
                n_local //= 2
        if n_local * world > 1:
            nt = e.table_count()
            mine = e.new_buffer(nt, n_local, 4)
            e.local_tables(mine)
            gathered = e.new_buffer(world, nt, n_local, 4)
            _all_gather(self.dist, self.group, gathered, mine, world)
            self.exchanges += 1
            full = gathered.permute(1, 2, 0, 3).contiguous()                  # entry j*world + g <- rank g, local j
            e.tail(full.view(nt, n_local * world, 4), n_local * world)
        if not collect:
            e.abort()
            return None
        return e.finish(finish_rounds if finish_rounds is not None else total_rounds)


_COMMIT_BUFS = {}


def sharded_commit(local_commit, sum_affine, world=1, group=None, dist=None, device=None):
    """KZG commit over (scalars, SRS) sharded across ranks.

    local_commit() -> (xy uint64[12], inf bool): this rank's sub-MSM;  sum_affine(xy [world,12], inf [world]) ->
    (xy, inf): group sum of the partial commitments.  Returns the full commitment on every rank.  The 104-byte records
    travel device to device (one pinned staging copy in, one all-gather, one copy out; the buffers are kept)."""
    import torch
    xy, inf = local_commit()
    key = (str(device), world)
    bufs = _COMMIT_BUFS.get(key)
    if bufs is None:
        pin = torch.empty(13, dtype=torch.int64)
        if device is not None and str(device).startswith("cuda"):
            pin = pin.pin_memory()
        bufs = _COMMIT_BUFS[key] = (pin, torch.empty(13, dtype=torch.int64, device=device), torch.empty((world, 13), dtype=torch.int64, device=device))
    pin, rec, out = bufs
    h = pin.numpy()
    h[:12] = np.ascontiguousarray(xy, dtype=np.uint64).view(np.int64)
    h[12] = 1 if inf else 0
    rec.copy_(pin, non_blocking=True)
    _all_gather(dist, group, out, rec, world)                            # C2: 104 bytes per rank
    g = out.cpu().numpy().view(np.uint64)
    return sum_affine(np.ascontiguousarray(g[:, :12]), np.ascontiguousarray(g[:, 12].astype(np.uint8)))


def hip_sum_affine(xy, inf):
    out = np.empty(12, dtype=np.uint64)
    oinf = C.c_uint8(0)
    N.check(N.lib().zkhip_g1_sum_affine(xy.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p),
                                        C.c_size_t(xy.shape[0]), out.ctypes.data_as(C.c_void_p), C.byref(oinf)), "g1_sum")
    return out, bool(oinf.value)

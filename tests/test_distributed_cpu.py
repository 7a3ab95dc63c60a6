"""world_size-2 (and 4) gloo tests of the multi-GPU exchange protocols on CPU.

The protocols live in the library as host logic independent of HIP (zk-cryptography_amd/csrc/shard_protocol.hpp: interleaved
sharding, stage / overlapped / round forms, per-exchange all-gather of partial sums, replicated transcript, gathered replicated tail).
Here the SAME C++ code is compiled for the host (tests/cpp/shard_protocol_host.cpp, tests/shard_host.py) and runs over gloo with
CHECKER engines built on the CPU oracle in place of the HIP engines; every rank must reproduce what the single-process oracle prover
yields on the full table.  (The commit merge -- one all-gather of partial commitments + a group sum -- needs the MSM and is covered on
the GPU: tests/test_gpu_two_ranks.py, tests/test_gpu_threads.py.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shard_host as H  # noqa: E402


class OracleSumcheckEngine:
    """Same split-phase interface as HipSumcheckEngine, computed by the oracle (test double)."""

    def __init__(self, ora, local_table, use_stages=True, overlap=False):
        self.ora = ora
        self.use_stages = use_stages
        self.overlap = overlap
        self.cur = np.ascontiguousarray(local_table, dtype=np.uint64)
        self.tr = ora.Transcript()
        self.rps, self.chs, self.sum = [], [], None
        self.cap = 8       # tiny "LDS capacity" so that both the collective rounds and the gathered tail are exercised

    def new_buffer(self, *shape):
        return torch.zeros(shape, dtype=torch.int64)

    def local_len(self):
        return self.cur.shape[0]

    def local_half_sums(self, out):
        out.copy_(torch.from_numpy(self.ora.mle_half_sums(self.cur).view(np.int64)))

    def _round(self, lo, hi, claimed_sum, first):
        o = self.ora
        if first:
            self.sum = np.asarray(claimed_sum, dtype=np.uint64) if claimed_sum is not None else o.fr_add(lo, hi)
            self.tr.commit(o.fr_to_bytes_be(self.sum))
        self.tr.commit(o.fr_to_bytes_be(lo) + o.fr_to_bytes_be(hi))
        r = self.tr.evaluate_challenge_into_field()
        self.rps.append(np.stack([lo, hi]))
        self.chs.append(r)
        return r

    def absorb(self, gathered, world, claimed_sum=None):
        g = gathered.numpy().view(np.uint64)
        lo, hi = g[0, 0].copy(), g[0, 1].copy()
        for k in range(1, world):
            lo, hi = self.ora.fr_add(lo, g[k, 0]), self.ora.fr_add(hi, g[k, 1])
        self._round(lo, hi, claimed_sum, not self.rps)

    def fold(self):
        self.cur = self.ora.mle_partial_evaluation(self.cur, self.chs[-1], 0)

    def local_value(self, out):
        out.copy_(torch.from_numpy(self.cur.view(np.int64)))

    def local_table(self, out):
        out.copy_(torch.from_numpy(self.cur.view(np.int64)))

    # ---- stage form: one exchange per k rounds
    # the plans are pure functions of (world, entries per shard): the protocol also calls them on a rank that has failed
    def stage_plan(self, world, n_local=None):
        n_local = self.cur.shape[0] if n_local is None else n_local
        n_glob = n_local * world
        if not self.use_stages or n_glob <= self.cap:
            return 0
        k = min(3, (n_glob // self.cap).bit_length() - 1, n_local.bit_length() - 1)
        self.k = k
        return k

    def stage_block_sums(self, out):
        m = self.cur.shape[0] >> self.k
        sums = np.stack([self.ora.mle_sum(self.cur[b * m:(b + 1) * m]) for b in range(1 << self.k)])
        out.copy_(torch.from_numpy(sums.view(np.int64)))

    def stage_absorb(self, gathered, world, claimed_sum=None):
        g = gathered.numpy().view(np.uint64)
        t = g[0].copy()
        for r in range(1, world):
            t = np.stack([self.ora.fr_add(t[b], g[r, b]) for b in range(t.shape[0])])
        self.stage_rs = []
        for _ in range(self.k):                      # k rounds on the block sums alone
            hs = self.ora.mle_half_sums(t)
            r = self._round(hs[0], hs[1], claimed_sum, not self.rps)
            t = self.ora.mle_partial_evaluation(t, r, 0)
            self.stage_rs.append(r)

    def stage_fold(self):
        for r in self.stage_rs:
            self.cur = self.ora.mle_partial_evaluation(self.cur, r, 0)

    # ---- overlapped stage: k1 rounds on coarse sums, k2 rounds on the block sums of the table folded by those k1 challenges
    # (two additive slices per rank, as the HIP engine's partial tables), leaving two local entries
    def _rounds_on(self, t, k, claimed_sum):
        rs = []
        for _ in range(k):
            hs = self.ora.mle_half_sums(t)
            r = self._round(hs[0], hs[1], claimed_sum, not self.rps)
            t = self.ora.mle_partial_evaluation(t, r, 0)
            rs.append(r)
        return rs

    def _block_sums(self, tab, blocks):
        m = tab.shape[0] // blocks
        return np.stack([self.ora.mle_sum(tab[b * m:(b + 1) * m]) for b in range(blocks)])

    def overlap_plan(self, world, n_local=None):
        L = (self.cur.shape[0] if n_local is None else n_local).bit_length() - 1
        if not self.overlap or L < 3:
            return None
        self.k1 = max(1, (L - 1) // 2)
        self.k2 = L - 1 - self.k1
        return self.k1, self.k2, 2 << self.k2

    def overlap_sums(self, out):
        out.copy_(torch.from_numpy(self._block_sums(self.cur, 1 << self.k1).view(np.int64)))

    def _sum_rows(self, rows):
        t = rows[0].copy()
        for r in range(1, rows.shape[0]):
            t = np.stack([self.ora.fr_add(t[b], rows[r, b]) for b in range(t.shape[0])])
        return t

    def overlap_rounds1(self, gathered, world, mid_out, claimed_sum=None):
        for r in self._rounds_on(self._sum_rows(gathered.numpy().view(np.uint64)), self.k1, claimed_sum):
            self.cur = self.ora.mle_partial_evaluation(self.cur, r, 0)
        blocks = 1 << self.k2
        m = self.cur.shape[0] // blocks                      # 2 entries per block: slice y takes entry y of every block
        mid = np.concatenate([np.stack([self.cur[b * m + y] for b in range(blocks)]) for y in range(2)])
        mid_out.copy_(torch.from_numpy(np.ascontiguousarray(mid).view(np.int64)))

    def overlap_rounds2(self, gathered, world):
        g = gathered.numpy().view(np.uint64).reshape(world * 2, 1 << self.k2, 4)
        for r in self._rounds_on(self._sum_rows(g), self.k2, None):
            self.cur = self.ora.mle_partial_evaluation(self.cur, r, 0)

    def tail_capacity(self):
        return self.cap

    def tail(self, values, m, claimed_sum=None):
        t = values.numpy().view(np.uint64).copy()
        while t.shape[0] > 1:
            hs = self.ora.mle_half_sums(t)
            r = self._round(hs[0], hs[1], claimed_sum, not self.rps)
            t = self.ora.mle_partial_evaluation(t, r, 0)

    def finish(self, n_rounds):
        assert len(self.rps) == n_rounds
        return self.sum, np.stack(self.rps), np.stack(self.chs)


def _worker(rank, world, port, log_n, q, so):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as ora
        from zk_cryptography_amd import distributed as D
        import shard_host as H
        full = ora.random_fr(1 << log_n, 4242)
        shard = D.shard_interleaved(full, rank, world)
        ws, wrp, wch = ora.sumcheck_prove(full)
        ok_sc = True
        for use_stages, overlap in ((True, False), (False, False), (True, True)):   # stage form, round form, overlapped stage
            eng = OracleSumcheckEngine(ora, shard, use_stages, overlap)
            (s, rp, ch), exchanges = H.prove_sumcheck(so, eng, world, dist)
            ok_sc = ok_sc and np.array_equal(s, ws) and np.array_equal(rp, wrp) and np.array_equal(ch, wch)
            if overlap and shard.shape[0] >= 8:
                ok_sc = ok_sc and exchanges == 3 and eng.k1 + eng.k2 + 1 + (world.bit_length() - 1) == log_n
        # a sum the caller claims is absorbed as given (and changes every challenge)
        five = ora.fr_from_ints([5])[0]
        (s5, rp5, ch5), _ = H.prove_sumcheck(so, OracleSumcheckEngine(ora, shard, True, True), world, dist, claimed_sum=five)
        ok_sc = ok_sc and np.array_equal(s5, five) and np.array_equal(rp5[0], wrp[0]) and (log_n < 2 or not np.array_equal(ch5, wch))

        ok_kzg = True      # (the commit merge is covered on the GPU)
        q.put((rank, bool(ok_sc), bool(ok_kzg)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n", [(2, 6), (4, 5), (2, 1)])
def test_sharded_protocol_gloo(world, log_n, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world * 3 + log_n
    so = H.build(tmp_path)
    procs = [ctx.Process(target=_worker, args=(r, world, port, log_n, q, so)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True, True) for r in range(world)]


class OracleComposedEngine:
    """Same split-phase interface as HipComposedEngine (zkhip_mc_*), computed by the oracle (test double).

    terms: list of uint64 [K_p, n_local, 4] arrays.  multi = False restates ComposedSumcheck::prove's round
    (composed_sumcheck.rs:41-57: raw evaluations absorbed), multi = True MultiComposedSumcheckProver::prove_internal's
    (multi_composed_sumcheck.rs:64-121: claimed sum first, per round the sum of the terms' interpolated polynomials)."""

    def __init__(self, ora, terms, multi, claimed_sum=None, cap=8):
        self.ora = ora
        self.terms = [np.ascontiguousarray(t, dtype=np.uint64) for t in terms]
        self.multi = multi
        self.tr = ora.Transcript()
        if multi:
            self.tr.commit(ora.fr_to_bytes_be(np.asarray(claimed_sum, dtype=np.uint64)))
        self.rps, self.chs = [], []
        self.cap = cap
        self.pending = None      # challenge the local tables are still to be folded at

    def new_buffer(self, *shape):
        return torch.zeros(shape, dtype=torch.int64)

    def record_len(self):
        return sum(t.shape[0] + 1 for t in self.terms)

    def table_count(self):
        return sum(t.shape[0] for t in self.terms)

    def tail_capacity(self):
        return self.cap

    def _apply_pending(self):
        if self.pending is not None:
            self.terms = [np.stack([self.ora.mle_partial_evaluation(t[k], self.pending, 0) for k in range(t.shape[0])]) for t in self.terms]
            self.pending = None

    def local_len(self):
        n = self.terms[0].shape[1]
        return n // 2 if self.pending is not None else n

    @staticmethod
    def _record(ora, terms):
        rec = []
        for t in terms:
            for x in range(t.shape[0] + 1):                      # evaluations at 0..=K of sum_j prod_k table_k(x, j)
                at = ora.fr_from_ints([x])[0]
                rec.append(ora.composed_sum(np.stack([ora.mle_partial_evaluation(t[k], at, 0) for k in range(t.shape[0])])))
        return np.stack(rec)

    def round_sums(self, out):
        self._apply_pending()
        out.copy_(torch.from_numpy(self._record(self.ora, self.terms).view(np.int64)))

    # ---- two rounds per exchange (zkhip_mc_stage_*): the record is, per term, C[a][b] = sum_j A[a][j] B[b][j] over the four blocks
    # of the next two variables (16 values, index 4 a + b) and 4 block sums of an additive table (none here: zeros)
    def stage_record_len(self, n_local=None):
        n_local = self.terms[0].shape[1] if n_local is None else n_local
        if n_local < 4 or any(t.shape[0] != 2 for t in self.terms):
            return 0
        assert self.pending is None or n_local != self.terms[0].shape[1]     # (no fold is pending inside the protocol's stage loop)
        return 20 * len(self.terms)

    def stage_sums(self, out):
        o, rec = self.ora, []
        zero = o.fr_from_ints([0])[0]
        for t in self.terms:
            m = t.shape[1] // 4
            for a in range(4):
                for b in range(4):
                    acc = zero
                    for j in range(m):
                        acc = o.fr_add(acc, o.fr_mul(t[0][a * m + j], t[1][b * m + j]))
                    rec.append(acc)
            rec += [zero] * 4
        out.copy_(torch.from_numpy(np.stack(rec).view(np.int64)))

    def stage_absorb(self, gathered, world):
        o = self.ora
        g = gathered.numpy().view(np.uint64)
        rec = g[0].copy()
        for k in range(1, world):
            rec = np.stack([o.fr_add(rec[i], g[k, i]) for i in range(rec.shape[0])])
        add, sub, mul = o.fr_add, o.fr_sub, o.fr_mul
        dbl = lambda v: add(v, v)           # noqa: E731
        one = o.fr_from_ints([1])[0]
        # round 1: p(0), p(1), p(2) of every term from C (the additive sums are zero here)
        evals = []
        for p_ in range(len(self.terms)):
            C = rec[20 * p_: 20 * p_ + 16]
            ll, hh = add(C[0], C[5]), add(C[10], C[15])
            lh = add(add(C[2], C[7]), add(C[8], C[13]))
            evals += [ll, hh, sub(add(ll, dbl(dbl(hh))), dbl(lh))]
        r1 = self._close(np.stack(evals))
        l0 = sub(one, r1)
        evals = []
        for p_ in range(len(self.terms)):
            C = rec[20 * p_: 20 * p_ + 16]
            B = []
            for x in range(2):
                for y in range(2):
                    t0 = add(mul(C[4 * x + y], l0), mul(C[4 * x + 2 + y], r1))
                    t1 = add(mul(C[4 * (2 + x) + y], l0), mul(C[4 * (2 + x) + 2 + y], r1))
                    B.append(add(mul(t0, l0), mul(t1, r1)))
            evals += [B[0], B[3], sub(add(B[0], dbl(dbl(B[3]))), dbl(add(B[1], B[2])))]
        r2 = self._close(np.stack(evals))
        for r in (r1, r2):
            self.terms = [np.stack([o.mle_partial_evaluation(t[k], r, 0) for k in range(t.shape[0])]) for t in self.terms]

    def _close(self, rec):
        o = self.ora
        if not self.multi:
            self.tr.commit(b"".join(o.fr_to_bytes_be(v) for v in rec))
            self.rps.append(rec.copy())
        else:
            rp, off = None, 0
            for t in self.terms:
                k = t.shape[0]
                term = o.sparse_interpolation(o.fr_from_ints(list(range(k + 1))), rec[off:off + k + 1])
                rp = term if rp is None else o.sparse_add(rp, term)
                off += k + 1
            if len(self.terms) == 1:
                rp = o.sparse_add(o.Sparse(), rp)                # zero() + term, as the reference's loop does
            self.tr.commit(o.sparse_to_bytes(rp))
            self.rps.append(rp.monomials())
        r = self.tr.evaluate_challenge_into_field()
        self.chs.append(r)
        return r

    def absorb(self, gathered, world):
        g = gathered.numpy().view(np.uint64)
        rec = g[0].copy()
        for k in range(1, world):
            rec = np.stack([self.ora.fr_add(rec[i], g[k, i]) for i in range(rec.shape[0])])
        self.pending = self._close(rec)

    def local_tables(self, out):
        self._apply_pending()
        out.copy_(torch.from_numpy(np.concatenate(self.terms).view(np.int64)))

    def tail(self, tables, m):
        t = tables.numpy().view(np.uint64).copy()
        terms, off = [], 0
        for old in self.terms:
            terms.append(t[off:off + old.shape[0]])
            off += old.shape[0]
        self.terms = terms
        while self.terms[0].shape[1] > 1:
            self.pending = self._close(self._record(self.ora, self.terms))
            self._apply_pending()

    def finish(self, n_rounds):
        assert len(self.rps) == n_rounds
        return (self.rps if self.multi else np.stack(self.rps)), np.stack(self.chs)


def _composed_worker(rank, world, port, log_n, q, so):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as ora
        from zk_cryptography_amd import distributed as D
        import shard_host as H
        n = 1 << log_n
        # ComposedSumcheck::prove, three tables
        full = np.stack([ora.random_fr(n, 77 + k) for k in range(3)])
        shard = np.stack([D.shard_interleaved(full[k], rank, world) for k in range(3)])
        (rp, ch), _ = H.prove_composed(so, OracleComposedEngine(ora, [shard], False), world, dist)
        wrp, wch = ora.composed_prove(full)
        ok_c = np.array_equal(rp, wrp) and np.array_equal(ch, wch)
        # MultiComposedSumcheckProver::prove_partial, the GKR shape (two terms of two tables) and a single term
        # ComposedSumcheck::prove with two tables: two rounds per exchange (stage records of 20 values)
        full2 = np.stack([ora.random_fr(n, 177 + k) for k in range(2)])
        shard2 = np.stack([D.shard_interleaved(full2[k], rank, world) for k in range(2)])
        (rp, ch), ex_stages = H.prove_composed(so, OracleComposedEngine(ora, [shard2], False), world, dist, use_stages=True)
        wrp, wch = ora.composed_prove(full2)
        ok_c = ok_c and np.array_equal(rp, wrp) and np.array_equal(ch, wch)
        (rp, ch), ex_plain = H.prove_composed(so, OracleComposedEngine(ora, [shard2], False), world, dist, use_stages=False)
        ok_c = ok_c and np.array_equal(rp, wrp) and np.array_equal(ch, wch)
        ok_c = ok_c and (ex_stages < ex_plain or n // world < 4)
        ok_m = True
        for sizes in ([2, 2], [3]):
            flat = np.stack([ora.random_fr(n, 91 + k) for k in range(sum(sizes))])
            s = ora.multi_composed_sum(flat, sizes)
            terms, off = [], 0
            for k in sizes:
                terms.append(np.stack([D.shard_interleaved(flat[off + i], rank, world) for i in range(k)]))
                off += k
            (rps, ch), _ = H.prove_composed(so, OracleComposedEngine(ora, terms, True, s), world, dist)
            orps, och = ora.multi_composed_prove(flat, sizes, s, partial=True)
            ok_m = ok_m and rps == [o.monomials() for o in orps] and np.array_equal(ch, och)
        q.put((rank, bool(ok_c), bool(ok_m)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n", [(2, 6), (4, 5), (2, 1)])
def test_sharded_composed_protocol_gloo(world, log_n, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world * 3 + log_n
    so = H.build(tmp_path)
    procs = [ctx.Process(target=_composed_worker, args=(r, world, port, log_n, q, so)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True, True) for r in range(world)]


# ---- a failing rank must not hang its peers (csrc/shard_protocol.hpp) ---------------------------------------------------------
def _failure_worker(rank, world, port, log_n, q, so):
    """Every protocol form, a failure injected on one rank in front of EVERY exchange index in turn: the failed rank returns its own
    status, every other rank ERR_PEER, all of them right behind the poisoned exchange -- and the next proof on the same group is right."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as ora
        from zk_cryptography_amd import distributed as D
        import shard_host as H
        NOMEM = -5
        n = 1 << log_n
        full = ora.random_fr(n, 515)
        shard = D.shard_interleaved(full, rank, world)
        want = ora.sumcheck_prove(full)
        full2 = np.stack([ora.random_fr(n, 616 + k) for k in range(2)])
        shard2 = np.stack([D.shard_interleaved(full2[k], rank, world) for k in range(2)])
        want2 = ora.composed_prove(full2)
        forms = [("sumcheck stages", lambda inj: H.prove_sumcheck(so, OracleSumcheckEngine(ora, shard, True, False), world, dist, inject=inj)),
                 ("sumcheck rounds", lambda inj: H.prove_sumcheck(so, OracleSumcheckEngine(ora, shard, False, False), world, dist, inject=inj)),
                 ("sumcheck overlapped", lambda inj: H.prove_sumcheck(so, OracleSumcheckEngine(ora, shard, True, True), world, dist, inject=inj)),
                 ("composed stages", lambda inj: H.prove_composed(so, OracleComposedEngine(ora, [shard2], False), world, dist, use_stages=True, inject=inj)),
                 ("composed rounds", lambda inj: H.prove_composed(so, OracleComposedEngine(ora, [shard2], False), world, dist, use_stages=False, inject=inj))]
        bad = []
        cases = 0
        for name, run in forms:
            _, n_ex = run(None)
            for idx in range(n_ex):
                failing = idx % world
                try:
                    run((idx, NOMEM) if rank == failing else None)
                    bad.append((name, idx, "no status"))
                except H.RankFailed as e:
                    if e.rc != (NOMEM if rank == failing else H.ERR_PEER) or e.exchanges != idx + 1:
                        bad.append((name, idx, e.rc, e.exchanges))
                cases += 1
            # the group is still in step: a healthy proof right behind the failures
            res, _ = run(None)
            w = want if name.startswith("sumcheck") else want2
            if not all(np.array_equal(a, b) for a, b in zip(res, w)):
                bad.append((name, "proof after failures differs"))
        q.put((rank, bad, cases))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n", [(2, 6), (4, 6)])
def test_failing_rank_does_not_hang_its_peers_gloo(world, log_n, tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world * 3 + log_n
    so = H.build(tmp_path)
    procs = [ctx.Process(target=_failure_worker, args=(r, world, port, log_n, q, so)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [b for _, b, _ in res] == [[]] * world, res
    assert all(c >= 10 for _, _, c in res)          # every exchange index of five protocol forms

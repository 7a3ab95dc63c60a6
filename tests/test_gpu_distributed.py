"""GPU checks of the sharded prover's HIP engine: (a) two shards driven in lockstep inside ONE process on one GPU
(the exchange done by stacking), (b) the real orchestration over torch.distributed/NCCL with world_size 1."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import zk_cryptography_amd as z
    return z


@pytest.mark.parametrize("world,log_n", [(2, 14), (4, 9), (2, 3), (8, 16), (2, 18), (8, 11)])   # (8, 11): the 2048-entry tail opens the transcript itself
def test_two_shards_in_lockstep_match_full_prover(zk, ora, world, log_n):
    import torch
    from zk_cryptography_amd import distributed as D
    full = ora.random_fr(1 << log_n, 99 + log_n)
    engines = []
    for g in range(world):
        t = torch.from_numpy(np.ascontiguousarray(D.shard_interleaved(full, g, world)).view(np.int64)).cuda()
        engines.append(D.HipSumcheckEngine(t))
    n_local = (1 << log_n) // world
    cap = engines[0].tail_capacity()
    sends = [e.new_buffer(2, 4) for e in engines]
    while n_local * world > cap and n_local > 1:
        for e, s in zip(engines, sends):
            e.local_half_sums(s)
        gathered = torch.stack(sends).contiguous()            # what the all-gather delivers to every rank
        for e in engines:
            e.absorb(gathered, world)
            e.fold()
        n_local //= 2
    tabs = []
    for e in engines:
        t = e.new_buffer(n_local, 4)
        e.local_table(t)
        tabs.append(t)
    rest = torch.stack(tabs).transpose(0, 1).contiguous().view(n_local * world, 4)
    outs = []
    for e in engines:
        e.tail(rest, n_local * world)
        outs.append(e.finish(log_n))
    ws, wrp, wch = ora.sumcheck_prove(full)
    for s, rp, ch in outs:
        assert np.array_equal(s, ws) and np.array_equal(rp, wrp) and np.array_equal(ch, wch)


def test_orchestration_over_rccl_world_1(zk, ora):
    """The in-library protocols over the library's OWN RCCL communicator (librccl resolved at run time; one rank -- RCCL refuses two
    ranks on one device): every exchange of the provers is a real ncclAllGather on the context's stream."""
    import torch
    from zk_cryptography_amd import _native as N
    from zk_cryptography_amd import distributed as D
    ctx = N.Context.get(0)
    comm = D.Comm(ctx, 1, 0, transport="rccl")
    try:
        for log_n in (13, 20):                     # stage form; overlapped stage (three exchanges, one beside the fold)
            full = ora.random_fr(1 << log_n, 5 + log_n)
            t = torch.from_numpy(full.view(np.int64)).cuda()
            sh = D.ShardedSumcheck(D.HipSumcheckEngine(t), 1, comm=comm)
            s, rp, ch = sh.prove()
            ws, wrp, wch = ora.sumcheck_prove(full)
            assert np.array_equal(s, ws) and np.array_equal(rp, wrp) and np.array_equal(ch, wch)
            assert sh.exchanges >= 2
        assert comm.stats()[0] >= 5                # they went through the communicator
        # composed prover, two rounds per exchange
        tabs = [ora.random_fr(1 << 15, 900 + q) for q in range(2)]
        shard = [torch.from_numpy(t_.view(np.int64)).cuda() for t_ in tabs]
        sh = D.ShardedComposedSumcheck(D.HipComposedEngine([shard], 1, multi=False), 1, use_stages=True, comm=comm)
        rp, ch = sh.prove()
        wrp, wch = ora.composed_prove(np.stack(tabs))
        assert np.array_equal(rp, wrp) and np.array_equal(ch, wch) and sh.exchanges > 0
        # sharded commit with one rank = plain commit
        tau = ora.random_fr(6, 3)
        srs = zk.TrustedSetup.setup(tau)
        poly = zk.Multilinear(ora.random_fr(64, 4))
        want = zk.MultilinearKZG.commitment(poly, srs)
        xy, inf = D.sharded_commit(srs.powers_of_tau_in_g1, srs.inf, poly.evaluations, comm)
        assert (not inf) and np.array_equal(xy, want.xy)
        # GKR, sharded entry point
        circuit = zk.Circuit.random(9)
        ev = circuit.evaluation(zk.Fr.random(2 ** 9, 309))
        want = zk.GKRProtocol.prove(circuit, ev)
        got = zk.GKRProtocol.prove_sharded(circuit, ev, 1, 0, use_stages=True, comm=comm)
        assert all(a.to_bytes() == b.to_bytes() for a, b in zip(got.sumcheck_proofs, want.sumcheck_proofs)) and got._exchanges > 0
        # the in-library cost of an exchange
        b2b, waited = comm.measure(64, 50)
        assert 0 < b2b <= waited * 1.5
    finally:
        comm.close()


def test_one_call_sharded_entry_points_world_1(zk, ora):
    """zkhip_sumcheck_prove_sharded / zkhip_composed_prove_sharded / zkhip_multi_composed_prove_sharded (begin + protocol + finish in
    one call) with a one-rank communicator without a transport."""
    import ctypes as C
    import torch
    from zk_cryptography_amd import _native as N
    from zk_cryptography_amd import distributed as D
    ctx = N.Context.get(0)
    comm = D.Comm.get(ctx, 1)
    lib = N.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    full = ora.random_fr(1 << 12, 77)
    t = torch.from_numpy(full.view(np.int64)).cuda()
    s, rp, ch = np.zeros(4, np.uint64), np.zeros((12, 2, 4), np.uint64), np.zeros((12, 4), np.uint64)
    ex = C.c_uint32(0)
    N.check(lib.zkhip_sumcheck_prove_sharded(comm.handle, N.ptr(t), C.c_size_t(1 << 12), None, p(s), p(rp), p(ch), C.byref(ex)), "sc")
    ws, wrp, wch = ora.sumcheck_prove(full)
    assert np.array_equal(s, ws) and np.array_equal(rp, wrp) and np.array_equal(ch, wch) and ex.value >= 1
    tabs = [ora.random_fr(1 << 12, 80 + q) for q in range(3)]
    dev = [torch.from_numpy(t_.view(np.int64)).cuda() for t_ in tabs]
    ptrs = (C.c_void_p * 3)(*[d.data_ptr() for d in dev])
    rp, ch = np.zeros((12, 4, 4), np.uint64), np.zeros((12, 4), np.uint64)
    N.check(lib.zkhip_composed_prove_sharded(comm.handle, ptrs, C.c_uint32(3), C.c_size_t(1 << 12), C.c_int(-1), p(rp), p(ch), C.byref(ex)), "composed")
    wrp, wch = ora.composed_prove(np.stack(tabs))
    assert np.array_equal(rp, wrp) and np.array_equal(ch, wch)
    sizes = [2, 1]
    claimed = ora.multi_composed_sum(np.stack(tabs), sizes)
    lens, rps, ch = np.zeros(12, np.uint32), np.zeros((12, 7, 2, 4), np.uint64), np.zeros((12, 4), np.uint64)
    N.check(lib.zkhip_multi_composed_prove_sharded(comm.handle, ptrs, (C.c_uint32 * 2)(*sizes), C.c_uint32(2), C.c_size_t(1 << 12), p(claimed),
                                                   C.c_int(-1), p(lens), p(rps), p(ch), C.byref(ex)), "multi")
    orps, och = ora.multi_composed_prove(np.stack(tabs), sizes, claimed, partial=True)
    got = [zk.SparseUnivariatePolynomial(rps[r, : lens[r], 0], rps[r, : lens[r], 1]).monomials() for r in range(12)]
    assert got == [o.monomials() for o in orps] and np.array_equal(ch, och)
    # shape errors as the reference panics, argument errors as statuses; a failed call leaves the context usable
    with pytest.raises(AssertionError):
        N.check(lib.zkhip_sumcheck_prove_sharded(comm.handle, N.ptr(t), C.c_size_t(3000), None, p(s), p(rp), p(ch), None), "sc")
    with pytest.raises(N.ZkhipError):
        N.check(lib.zkhip_sumcheck_prove_sharded(None, N.ptr(t), C.c_size_t(1 << 12), None, p(s), p(rp), p(ch), None), "sc")
    with pytest.raises(AssertionError):
        D.Comm(ctx, 3, 0, transport="staged")
    s2, rp2, ch2 = np.zeros(4, np.uint64), np.zeros((12, 2, 4), np.uint64), np.zeros((12, 4), np.uint64)
    N.check(lib.zkhip_sumcheck_prove_sharded(comm.handle, N.ptr(t), C.c_size_t(1 << 12), None, p(s2), p(rp2), p(ch2), None), "sc")
    want = ora.sumcheck_prove(full)
    assert np.array_equal(s2, want[0]) and np.array_equal(rp2, want[1]) and np.array_equal(ch2, want[2])


@pytest.mark.parametrize("world,log_n", [(2, 14), (4, 13), (8, 20), (2, 22), (8, 12), (4, 5), (2, 19)])   # (2, 19): one 9-variable stage
def test_stage_form_shards_in_lockstep_match_full_prover(zk, ora, world, log_n):
    """Stage form (one exchange per k rounds) with `world` shards driven in lockstep on one GPU."""
    import torch
    from zk_cryptography_amd import distributed as D
    full = ora.random_fr(1 << log_n, 199 + log_n)
    engines = []
    for g in range(world):
        t = torch.from_numpy(np.ascontiguousarray(D.shard_interleaved(full, g, world)).view(np.int64)).cuda()
        engines.append(D.HipSumcheckEngine(t))
    n_local = (1 << log_n) // world
    n_exchanges = 0
    while True:
        ks = [e.stage_plan(world) for e in engines]
        assert len(set(ks)) == 1
        k = ks[0]
        if k == 0:
            break
        mine = []
        for e in engines:
            b = e.new_buffer(1 << k, 4)
            e.stage_block_sums(b)
            mine.append(b)
        gathered = torch.stack(mine).contiguous()
        for e in engines:
            e.stage_absorb(gathered, world)
            e.stage_fold()
        n_local >>= k
        n_exchanges += 1
    tabs = []
    for e in engines:
        t = e.new_buffer(n_local, 4)
        e.local_table(t)
        tabs.append(t)
    rest = torch.stack(tabs).transpose(0, 1).contiguous().view(n_local * world, 4)
    outs = []
    for e in engines:
        e.tail(rest, n_local * world)
        outs.append(e.finish(log_n))
    ws, wrp, wch = ora.sumcheck_prove(full)
    for s, rp, ch in outs:
        assert np.array_equal(s, ws) and np.array_equal(rp, wrp) and np.array_equal(ch, wch)
    assert n_exchanges <= 3


def test_stage_form_2_27_over_8_shards_matches_single_gpu_prover(zk):
    """The 8-GPU bench shape (2^24 entries per rank): 8 + 8 variables in two stages, then the gathered 2^11-entry tail --
    three exchanges.  Eight shards driven in lockstep on one GPU against the single-GPU prover on the full table
    (itself checked against the oracle up to 2^24)."""
    import torch
    from zk_cryptography_amd import distributed as D
    world, log_n = 8, 27
    g = torch.Generator(device="cuda").manual_seed(27)
    full = torch.randint(0, 2 ** 62, (1 << log_n, 4), dtype=torch.int64, device="cuda", generator=g)
    sc = zk.Sumcheck(zk.Multilinear(full))
    sc.poly_sum()
    want, want_ch = sc.prove()
    engines = [D.HipSumcheckEngine(full[r::world].contiguous()) for r in range(world)]
    n_local, ks = (1 << log_n) // world, []
    while True:
        k = engines[0].stage_plan(world)
        assert all(e.stage_plan(world) == k for e in engines[1:])
        if k == 0:
            break
        mine = []
        for e in engines:
            b = e.new_buffer(1 << k, 4)
            e.stage_block_sums(b)
            mine.append(b)
        gathered = torch.stack(mine).contiguous()
        for e in engines:
            e.stage_absorb(gathered, world)
            e.stage_fold()
        n_local >>= k
        ks.append(k)
    assert ks == [8, 8] and n_local * world == 2048
    tabs = []
    for e in engines:
        t = e.new_buffer(n_local, 4)
        e.local_table(t)
        tabs.append(t)
    rest = torch.stack(tabs).transpose(0, 1).contiguous().view(n_local * world, 4)
    for e in engines:
        e.tail(rest, n_local * world)
        s, rp, ch = e.finish(log_n)
        assert np.array_equal(s, want.sum) and np.array_equal(rp, want.univariate_poly) and np.array_equal(ch, want_ch)


def test_gathered_tail_of_2048_entries_with_a_claimed_sum(zk, ora):
    """zkhip_sc_tail on a 2048-entry table that no round has touched yet: its first round (run by itself: the serial kernel holds 1024
    entries) opens the transcript and absorbs the caller's sum as given."""
    import torch
    from zk_cryptography_amd import distributed as D
    full = ora.random_fr(1 << 11, 4711)
    t = torch.from_numpy(full.view(np.int64)).cuda()
    five = zk.Fr.from_int(5)
    for claimed in (None, five):
        sh = D.ShardedSumcheck(D.HipSumcheckEngine(t), 1, None, None)
        s, rp, ch = sh.prove(claimed_sum=claimed)
        sc = zk.Sumcheck(zk.Multilinear(full))
        if claimed is None:
            sc.poly_sum()
        else:
            sc.sum = claimed
        want, wch = sc.prove()
        assert np.array_equal(s, want.sum) and np.array_equal(rp, want.univariate_poly) and np.array_equal(ch, wch)
    ws, wrp, wch = ora.sumcheck_prove(full)
    sh = D.ShardedSumcheck(D.HipSumcheckEngine(t), 1, None, None)
    s, rp, ch = sh.prove()
    assert np.array_equal(s, ws) and np.array_equal(rp, wrp) and np.array_equal(ch, wch)


def _drive_overlapped(torch, engines, world, log_n):
    """The overlapped stage with `world` shards in lockstep on one GPU; returns every engine's proof."""
    plans = [e.overlap_plan(world) for e in engines]
    assert plans[0] is not None and len(set(plans)) == 1
    k1, k2, mid = plans[0]
    mine = []
    for e in engines:
        b = e.new_buffer(1 << k1, 4)
        e.overlap_sums(b)
        mine.append(b)
    gathered = torch.stack(mine).contiguous()
    mids = []
    for e in engines:
        b = e.new_buffer(mid, 4)
        e.overlap_rounds1(gathered, world, b)
        mids.append(b)
    gathered = torch.stack(mids).contiguous()
    for e in engines:
        e.overlap_rounds2(gathered, world)
    assert all(e.local_len() == 256 for e in engines) and all(e.stage_plan(world) == 0 for e in engines)
    assert k1 + k2 + 8 + world.bit_length() - 1 == log_n and 256 * world <= engines[0].tail_capacity()
    tabs = []
    for e in engines:
        t = e.new_buffer(256, 4)
        e.local_table(t)
        tabs.append(t)
    rest = torch.stack(tabs).transpose(0, 1).contiguous().view(256 * world, 4)
    outs = []
    for e in engines:
        e.tail(rest, 256 * world)
        outs.append(e.finish(log_n))
    return outs


@pytest.mark.parametrize("world,log_n", [(1, 19), (2, 20), (4, 21), (8, 22), (2, 22), (4, 24)])
def test_overlapped_stage_shards_in_lockstep_match_full_prover(zk, ora, world, log_n):
    """The overlapped stage (shards of 2^19..2^24 entries: rounds on coarse sums, then on the folded fine sums beside the
    shard's fold) against the oracle on the full table; (8, 22) ends in the 2048-entry tail."""
    import torch
    from zk_cryptography_amd import distributed as D
    full = ora.random_fr(1 << log_n, 299 + log_n)
    engines = []
    for g in range(world):
        t = torch.from_numpy(np.ascontiguousarray(D.shard_interleaved(full, g, world)).view(np.int64)).cuda()
        engines.append(D.HipSumcheckEngine(t))
    outs = _drive_overlapped(torch, engines, world, log_n)
    ws, wrp, wch = ora.sumcheck_prove(full)
    for s, rp, ch in outs:
        assert np.array_equal(s, ws) and np.array_equal(rp, wrp) and np.array_equal(ch, wch)
    # the protocol driver picks the same path (world 1: the exchange is a copy)
    if world == 1:
        sh = D.ShardedSumcheck(D.HipSumcheckEngine(torch.from_numpy(full.view(np.int64)).cuda()), 1, None, None)
        s, rp, ch = sh.prove()
        assert sh.exchanges == 3 and np.array_equal(s, ws) and np.array_equal(rp, wrp) and np.array_equal(ch, wch)
        # a claimed sum the caller overrides is absorbed as given, as by the single-GPU prover
        five = zk.Fr.from_int(5)
        sh = D.ShardedSumcheck(D.HipSumcheckEngine(torch.from_numpy(full.view(np.int64)).cuda()), 1, None, None)
        s5, rp5, ch5 = sh.prove(claimed_sum=five)
        sc = zk.Sumcheck(zk.Multilinear(full))
        sc.sum = five
        want5, wch5 = sc.prove()
        assert np.array_equal(s5, want5.sum) and np.array_equal(rp5, want5.univariate_poly) and np.array_equal(ch5, wch5)
        assert not np.array_equal(ch5, wch)
    # shards outside the plan's range fall back to the stage form
    small = D.HipSumcheckEngine(torch.from_numpy(np.ascontiguousarray(full[: 1 << 12]).view(np.int64)).cuda())
    assert small.overlap_plan(world) is None
    small.abort()


def test_overlapped_stage_2_27_over_8_shards_matches_single_gpu_prover(zk):
    """The 8-GPU bench shape (2^24 entries per rank) on the overlapped stage: 6 rounds | 10 rounds beside the 6-variable fold
    | 2048-entry gathered tail, three exchanges.  Eight shards in lockstep on one GPU against the single-GPU prover."""
    import torch
    from zk_cryptography_amd import distributed as D
    world, log_n = 8, 27
    g = torch.Generator(device="cuda").manual_seed(2727)
    full = torch.randint(0, 2 ** 62, (1 << log_n, 4), dtype=torch.int64, device="cuda", generator=g)
    sc = zk.Sumcheck(zk.Multilinear(full))
    sc.poly_sum()
    want, want_ch = sc.prove()
    engines = [D.HipSumcheckEngine(full[r::world].contiguous()) for r in range(world)]
    assert engines[0].overlap_plan(world)[:2] == (6, 10)
    for s, rp, ch in _drive_overlapped(torch, engines, world, log_n):
        assert np.array_equal(s, want.sum) and np.array_equal(rp, want.univariate_poly) and np.array_equal(ch, want_ch)


def _drive_composed_shards(D, torch, engines, world, n_local, stages=False, counts=None):
    """What ShardedComposedSumcheck.prove does, with the all-gather done by stacking the shards' buffers.  stages: two rounds per
    exchange where the claim allows it (zkhip_mc_stage_*: every term a product of two tables)."""
    cap, rec, nt = engines[0].tail_capacity(), engines[0].record_len(), engines[0].table_count()
    sends = [e.new_buffer(rec, 4) for e in engines]
    n_exchanges = n_rounds_done = 0
    while stages and n_local * world > cap and n_local >= 4:
        vals = engines[0].stage_record_len()
        assert all(e.stage_record_len() == vals for e in engines)
        if not vals:
            break
        st_sends = [e.new_buffer(vals, 4) for e in engines]
        for e, s in zip(engines, st_sends):
            e.stage_sums(s)
        gathered = torch.stack(st_sends).contiguous()
        for e in engines:
            e.stage_absorb(gathered, world)
        n_local //= 4
        n_exchanges += 1
        n_rounds_done += 2
    if counts is not None:
        counts["stage_exchanges"] = n_exchanges
    n_stage_exchanges = n_exchanges
    while n_local * world > cap and n_local > 1:
        for e, s in zip(engines, sends):
            e.round_sums(s)
        gathered = torch.stack(sends).contiguous()
        for e in engines:
            e.absorb(gathered, world)
        n_local //= 2
        n_exchanges += 1
    tabs = []
    for e in engines:
        assert e.local_len() == n_local
        t = e.new_buffer(nt, n_local, 4)
        e.local_tables(t)
        tabs.append(t)
    full = torch.stack(tabs).permute(1, 2, 0, 3).contiguous().view(nt, n_local * world, 4)
    total_rounds = (n_local * world).bit_length() - 1 + (n_exchanges - n_stage_exchanges) + 2 * n_stage_exchanges
    for e in engines:
        e.tail(full, n_local * world)
    return [e.finish(total_rounds) for e in engines]


@pytest.mark.parametrize("world,k,log_n", [(2, 2, 14), (4, 3, 13), (8, 5, 12), (2, 1, 11), (8, 2, 4), (2, 2, 1), (4, 2, 18)])
def test_sharded_composed_sumcheck_matches_full_prover(zk, ora, world, k, log_n):
    """ComposedSumcheck::prove with every table split over `world` shards driven in lockstep on one GPU: round polynomials
    and challenges must be those of the oracle (and so of the single-GPU prover) on the whole tables."""
    import torch
    from zk_cryptography_amd import distributed as D
    from zk_cryptography_amd import _native as N
    full = np.stack([ora.random_fr(1 << log_n, 300 + 7 * q + log_n) for q in range(k)])
    engines = []
    for g in range(world):
        shard = [torch.from_numpy(np.ascontiguousarray(D.shard_interleaved(full[q], g, world)).view(np.int64)).cuda() for q in range(k)]
        engines.append(D.HipComposedEngine([shard], world, multi=False, ctx=N.Context(0)))
    outs = _drive_composed_shards(D, torch, engines, world, (1 << log_n) // world)
    wrp, wch = ora.composed_prove(full)
    for rp, ch in outs:
        assert np.array_equal(rp, wrp) and np.array_equal(ch, wch)
    if k == 2:     # the same proof with two rounds per exchange
        engines = []
        for g in range(world):
            shard = [torch.from_numpy(np.ascontiguousarray(D.shard_interleaved(full[q], g, world)).view(np.int64)).cuda() for q in range(k)]
            engines.append(D.HipComposedEngine([shard], world, multi=False, ctx=N.Context(0)))
        counts = {}
        outs = _drive_composed_shards(D, torch, engines, world, (1 << log_n) // world, stages=True, counts=counts)
        for rp, ch in outs:
            assert np.array_equal(rp, wrp) and np.array_equal(ch, wch)
        assert counts["stage_exchanges"] > 0 or (1 << log_n) // world < 4 or (1 << log_n) <= engines[0].tail_capacity()


@pytest.mark.parametrize("world,sizes,log_n", [(2, [2, 2], 12), (8, [2, 3], 13), (4, [1, 5], 10), (2, [2, 2, 1, 1], 7), (4, [3], 16)])
def test_sharded_multi_composed_prove_partial_matches_full_prover(zk, ora, world, sizes, log_n):
    """MultiComposedSumcheckProver::prove_partial (what GKR calls per layer) over sharded tables."""
    import torch
    from zk_cryptography_amd import distributed as D
    from zk_cryptography_amd import _native as N
    flat = np.stack([ora.random_fr(1 << log_n, 500 + 3 * q + log_n) for q in range(sum(sizes))])
    s = ora.multi_composed_sum(flat, sizes)
    engines = []
    for g in range(world):
        terms, q = [], 0
        for k in sizes:
            terms.append([torch.from_numpy(np.ascontiguousarray(D.shard_interleaved(flat[q + i], g, world)).view(np.int64)).cuda()
                          for i in range(k)])
            q += k
        engines.append(D.HipComposedEngine(terms, world, multi=True, claimed_sum=s, ctx=N.Context(0)))
    stages = all(k == 2 for k in sizes)          # the GKR shape: two rounds per exchange
    counts = {}
    outs = _drive_composed_shards(D, torch, engines, world, (1 << log_n) // world, stages=stages, counts=counts)
    orps, och = ora.multi_composed_prove(flat, sizes, s, partial=True)
    want = [o.monomials() for o in orps]
    for rps, ch in outs:
        got = [zk.SparseUnivariatePolynomial(c, p).monomials() for c, p in rps]
        assert got == want and np.array_equal(ch, och)
    assert not stages or counts["stage_exchanges"] > 0


def test_sharded_composed_orchestration_world_1(zk, ora):
    """ShardedComposedSumcheck end to end with one rank (no process group needed: the gather is a copy)."""
    import torch
    from zk_cryptography_amd import distributed as D
    full = np.stack([ora.random_fr(1 << 15, 900 + q) for q in range(3)])
    shard = [torch.from_numpy(full[q].view(np.int64)).cuda() for q in range(3)]
    rp, ch = D.ShardedComposedSumcheck(D.HipComposedEngine([shard], 1, multi=False), 1).prove()
    wrp, wch = ora.composed_prove(full)
    assert np.array_equal(rp, wrp) and np.array_equal(ch, wch)


def test_sharded_composed_session_misuse_is_refused(zk, ora):
    """The split-phase session checks its call order and shapes (include/zkhip.h, zkhip_mc_*)."""
    import ctypes as C
    import torch
    from zk_cryptography_amd import _native as N
    from zk_cryptography_amd import distributed as D
    t = [torch.from_numpy(ora.random_fr(1 << 13, 1 + q).view(np.int64)).cuda() for q in range(2)]
    with pytest.raises(AssertionError):                       # world must be a power of two
        D.HipComposedEngine([t], 3, multi=False)
    with pytest.raises(N.ZkhipError):                         # a multi-composed claim needs its sum
        D.HipComposedEngine([t], 2, multi=True)
    with pytest.raises(N.ZkhipError):                         # ComposedSumcheck is one term
        D.HipComposedEngine([t, t], 2, multi=False)
    e = D.HipComposedEngine([t], 2, multi=False)
    rec = e.new_buffer(e.record_len(), 4)
    gathered = e.new_buffer(2, e.record_len(), 4)
    with pytest.raises(N.ZkhipError):                         # nothing to absorb yet
        e.absorb(gathered, 2)
    e.round_sums(rec)
    with pytest.raises(N.ZkhipError):                         # the round is still open
        e.round_sums(rec)
    with pytest.raises(N.ZkhipError):                         # other world size than at begin
        e.absorb(e.new_buffer(4, e.record_len(), 4), 4)
    with pytest.raises(N.ZkhipError):                         # tables are gathered between rounds only
        e.local_tables(e.new_buffer(e.table_count(), e.local_len(), 4))
    gathered.zero_()
    gathered[0] = rec
    e.absorb(gathered, 2)
    with pytest.raises(AssertionError):                       # the remaining tables do not fit the replicated tail yet
        e.tail(e.new_buffer(e.table_count(), 1 << 13, 4), 1 << 13)
    # release without collecting a proof
    N.check(N.lib().zkhip_mc_finish(e.st, None, None, None), "mc_finish")
    e.st = None

"""The threading contract of the C ABI (include/zkhip.h: "one context per host thread; contexts are independent" -- the reference's
types are plain data, trivially Send, polynomial/src/multilinear/evaluation_form.rs:6-9) and the resource contract behind it:
  * 2 and 4 host threads, one zkhip_ctx and one HIP stream each, prove + commit + open + NTT at the same time, every result against
    the oracle;
  * threads as RANKS: the in-library sharded provers (zkhip_sc_prove_sharded / zkhip_mc_prove_sharded / zkhip_gkr_prove_sharded /
    zkhip_kzg_commit_sharded) with a callback communicator whose all-gather is a barrier + a shared host buffer -- real concurrent
    ranks with real exchanges on one GPU, in one process;
  * a soak: thousands of create / destroy, begin / abort and begin / end cycles give all device memory back."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import zk_cryptography_amd as z
    return z


def _run_threads(n, body):
    """body(i) on n threads at once; re-raises the first exception"""
    errors, barrier = [], threading.Barrier(n)

    def run(i):
        try:
            barrier.wait(timeout=60)
            body(i)
        except BaseException as e:      # noqa: BLE001
            errors.append(e)
            barrier.abort()

    ts = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ts), "a thread hung"
    if errors:
        raise errors[0]


@pytest.mark.parametrize("n_threads", [2, 4])
def test_contexts_on_concurrent_host_threads(zk, ora, n_threads):
    import torch
    from zk_cryptography_amd import _native as N
    log_sc, log_kzg, log_open = 16, 9, 7
    # inputs and expected values per thread, from the oracle, before the threads start
    cases = []
    for i in range(n_threads):
        ev = ora.random_fr(1 << log_sc, 7000 + i)
        tau = ora.random_fr(log_kzg, 7100 + i)
        sc = ora.random_fr(1 << log_kzg, 7200 + i)
        srs_j = ora.kzg_multilinear_srs_g1(tau)
        tau_o = ora.random_fr(log_open, 7300 + i)
        ev_o = ora.random_fr(1 << log_open, 7400 + i)
        z = ora.random_fr(log_open, 7500 + i)
        w_eval, w_jac = ora.kzg_open(ev_o, z, ora.kzg_multilinear_srs_g1(tau_o))
        cases.append(dict(ev=ev, want_sc=ora.sumcheck_prove(ev), tau=tau, sc=sc, want_commit=ora.g1_to_affine(ora.kzg_commitment(sc, srs_j, True)),
                          tau_o=tau_o, ev_o=ev_o, z=z, want_open=(w_eval, [ora.g1_to_affine(pj) for pj in w_jac]),
                          a=ora.random_fr(1 << 12, 7600 + i)))
    handles = [None] * n_threads

    def body(i):
        c = cases[i]
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx = N.Context.get(0)                    # this thread's own context, on this thread's stream
            handles[i] = ctx.handle.value
            for rep in range(3):
                poly = zk.Multilinear(c["ev"])
                s = zk.Sumcheck(poly)
                s.poly_sum()
                proof, ch = s.prove()
                ws, wrp, wch = c["want_sc"]
                assert np.array_equal(proof.sum, ws) and np.array_equal(proof.univariate_poly, wrp) and np.array_equal(ch, wch)
                srs = zk.TrustedSetup.setup(c["tau"])
                got = zk.MultilinearKZG.commitment(zk.Multilinear(c["sc"]), srs)
                assert (not got.infinity) and np.array_equal(got.xy, c["want_commit"][:12])
                srs_o = zk.TrustedSetup.setup(c["tau_o"])
                op = zk.MultilinearKZG.open(zk.Multilinear(c["ev_o"]), c["z"], srs_o)
                w_eval, w_proofs = c["want_open"]
                assert np.array_equal(op.evaluation, w_eval)
                for k, p in enumerate(op.proofs):
                    assert p.infinity == bool(w_proofs[k][12]) and (p.infinity or np.array_equal(p.xy, w_proofs[k][:12]))
                dom = zk.Domain(1 << 12)
                back = dom.ifft(dom.fft(c["a"]))
                assert np.array_equal(back.cpu().numpy().view(np.uint64), c["a"])
            stream.synchronize()

    _run_threads(n_threads, body)
    assert len(set(handles)) == n_threads             # one context per thread


class BarrierExchange:
    """all-gather for `world` threads of one process: every rank copies its payload into a shared host buffer, a barrier, every rank
    reads the whole buffer.  (zkhip_memcpy_d2h / _h2d wait for the context's stream, as the callback contract allows.)"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.buf = np.zeros(world * (1 << 20), dtype=np.uint8)

    def callback(self, ctx, rank):
        from zk_cryptography_amd import _native as N
        from zk_cryptography_amd import distributed as D
        lib, world, buf, barrier = N.lib(), self.world, self.buf, self.barrier
        err = []

        def fn(user, d_send, d_recv, nbytes, stream):
            try:
                assert nbytes * world <= buf.size
                mine = buf[rank * nbytes:(rank + 1) * nbytes]
                N.check(lib.zkhip_memcpy_d2h(ctx.handle, mine.ctypes.data_as(C.c_void_p), C.c_void_p(d_send), C.c_size_t(nbytes)), "d2h")
                barrier.wait(timeout=120)
                N.check(lib.zkhip_memcpy_h2d(ctx.handle, C.c_void_p(d_recv), buf.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes * world)), "h2d")
                barrier.wait(timeout=120)      # nobody overwrites the buffer before everyone has read it
                return 0
            except BaseException as e:         # noqa: BLE001
                err.append(e)
                barrier.abort()
                return 1
        return D.ALL_GATHER_FN(fn), err


@pytest.mark.parametrize("world", [2, 4])
def test_threads_as_ranks_run_the_in_library_sharded_provers(zk, ora, world):
    import torch
    from zk_cryptography_amd import _native as N
    from zk_cryptography_amd import distributed as D
    xch = BarrierExchange(world)
    full = zk.Fr.random(1 << 21, 5150)                                   # shards of 2^20 / 2^19: the overlapped stage
    sc = zk.Sumcheck(zk.Multilinear(full))
    sc.poly_sum()
    want_sc, want_ch = sc.prove()
    tabs = [zk.Fr.random(1 << 15, 5160 + k) for k in range(4)]
    poly = [zk.ComposedMultilinear([zk.Multilinear(tabs[0]), zk.Multilinear(tabs[1])]),
            zk.ComposedMultilinear([zk.Multilinear(tabs[2]), zk.Multilinear(tabs[3])])]
    claimed = zk.MultiComposedSumcheckProver.calculate_poly_sum(poly)
    want_mc, want_mc_ch = zk.MultiComposedSumcheckProver.prove_partial(poly, claimed)
    want_mc = [p.monomials() for p in want_mc.round_polys]
    circuit = zk.Circuit.random(12)
    gkr_in = zk.Fr.random(2 ** 12, 5170)
    want_gkr = zk.GKRProtocol.prove(circuit, circuit.evaluation(gkr_in))
    tau = zk.Fr.random(11, 5180)
    srs = zk.TrustedSetup.setup(tau)
    scal = zk.Fr.random(1 << 11, 5181)
    want_c = zk.MultilinearKZG.commitment(zk.Multilinear(scal), srs)
    srs_xy, srs_inf = srs.powers_of_tau_in_g1.cpu(), srs.inf.cpu()
    torch.cuda.synchronize()
    exchanges = [None] * world

    def body(rank):
        with torch.cuda.stream(torch.cuda.Stream()):
            ctx = N.Context.get(0)
            cb, err = xch.callback(ctx, rank)
            comm = D.Comm(ctx, world, rank, transport=cb)              # a comm over the thread exchange (not a torch.distributed one)

            def cuda(a):
                return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()

            try:
                s, rp, ch = D.ShardedSumcheck(D.HipSumcheckEngine(cuda(full[rank::world])), world, comm=comm).prove()
                assert np.array_equal(s, want_sc.sum) and np.array_equal(rp, want_sc.univariate_poly) and np.array_equal(ch, want_ch)
                sh = [cuda(t[rank::world]) for t in tabs]
                eng = D.HipComposedEngine([sh[:2], sh[2:]], world, multi=True, claimed_sum=claimed)
                rps, ch = D.ShardedComposedSumcheck(eng, world, comm=comm).prove()
                assert [zk.SparseUnivariatePolynomial(c_, p_).monomials() for c_, p_ in rps] == want_mc and np.array_equal(ch, want_mc_ch)
                my_circuit = zk.Circuit.random(12)                          # every rank holds the circuit and the layer values
                got = zk.GKRProtocol.prove_sharded(my_circuit, my_circuit.evaluation(gkr_in), world, rank, comm=comm)
                assert all(a.to_bytes() == b.to_bytes() for a, b in zip(got.sumcheck_proofs, want_gkr.sumcheck_proofs))
                assert all(np.array_equal(a, b) for a, b in zip(got.wb_s, want_gkr.wb_s)) and all(np.array_equal(a, b) for a, b in zip(got.wc_s, want_gkr.wc_s))
                xy, inf = D.sharded_commit(srs_xy[rank::world].contiguous().cuda(), srs_inf[rank::world].contiguous().cuda(), cuda(scal[rank::world]), comm)
                assert (not inf) and np.array_equal(xy, want_c.xy)
                exchanges[rank] = comm.stats()[0]
            finally:
                if err:
                    raise err[0]
                comm.close()

    _run_threads(world, body)
    assert len(set(exchanges)) == 1 and exchanges[0] > 10                  # every rank took part in every exchange


@pytest.mark.parametrize("n_threads", [4])
def test_threads_prove_one_circuit_at_once_each_on_its_own_stream(zk, ora, n_threads):
    """Several host threads, a context and a stream each, prove ONE Circuit object at the same time (how a prover gets GKR throughput:
    tools/gkr_threads.py): the mirror keeps a device copy of the circuit per context, every proof is the synchronous one."""
    import torch
    depth = 7
    circuit = zk.Circuit.random(depth)
    inp = ora.random_fr(2 ** depth, 9100)
    ev = circuit.evaluation(inp)
    want = [p.to_bytes() for p in zk.GKRProtocol.prove(circuit, ev).sumcheck_proofs]
    got, errors, gate = [None] * n_threads, [], threading.Barrier(n_threads)

    def work(k):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                gate.wait()
                for _ in range(6):
                    pr = zk.GKRProtocol.prove(circuit, ev)
                    assert [p.to_bytes() for p in pr.sumcheck_proofs] == want
                got[k] = True
        except BaseException as e:   # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))
    ths = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors and all(got), errors
    # the main thread's copy and one per worker context -- whose contexts (and device copies) go with their threads: the Circuit keeps none of
    # them alive, and a worker that arrives late already drops the copies of those that have finished
    assert 1 <= len(circuit._devices) <= n_threads + 1 and sum(d.alive() for d in circuit._devices) >= 1
    zk.GKRProtocol.prove(circuit, ev)                        # the next proof drops the dead entries
    assert all(d.alive() for d in circuit._devices)


def test_soak_create_destroy_and_aborted_sessions_return_all_memory(zk, ora):
    """2000 cycles each of context create / destroy (with work in between), zkhip_sc_begin / _abort, zkhip_mc_begin / _abort,
    commit_begin / _end and comm create / destroy: free device memory before = after (hipMemGetInfo), and the context still proves."""
    import torch
    from zk_cryptography_amd import _native as N
    from zk_cryptography_amd import distributed as D
    lib = N.lib()
    ev = ora.random_fr(1 << 12, 9100)
    t = torch.from_numpy(ev.view(np.int64)).cuda()
    tabs = [torch.from_numpy(ora.random_fr(1 << 12, 9101 + q).view(np.int64)).cuda() for q in range(2)]
    srs = zk.TrustedSetup.setup(ora.random_fr(10, 9110))
    sc = torch.from_numpy(ora.random_fr(1 << 10, 9111).view(np.int64)).cuda()
    out_s, out_rp, out_ch = np.zeros(4, np.uint64), np.zeros((12, 2, 4), np.uint64), np.zeros((12, 4), np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731

    def cycle(n_ctx, n_sessions):
        for k in range(n_ctx):
            h = C.c_void_p()
            N.check(lib.zkhip_ctx_create(C.byref(h), C.c_int(0), None), "ctx_create")
            if k % 50 == 0:       # some contexts do real work first: provers, a commit in flight that is never collected, an NTT plan
                N.check(lib.zkhip_sumcheck_prove(h, N.ptr(t), C.c_size_t(1 << 12), None, None, None, C.c_uint32(0), p(out_s), p(out_rp), p(out_ch)), "prove")
                buf = t.clone()
                N.check(lib.zkhip_ntt(h, N.ptr(buf), C.c_uint32(12), C.c_int(0)), "ntt")
                m = C.c_void_p()
                N.check(lib.zkhip_comm_create(h, C.c_uint32(0), C.c_uint32(1), None, None, C.byref(m)), "comm")
                N.check(lib.zkhip_sumcheck_prove_sharded(m, N.ptr(t), C.c_size_t(1 << 12), None, p(out_s), p(out_rp), p(out_ch), None), "sharded")
                N.check(lib.zkhip_comm_destroy(m), "comm_destroy")
                tk = C.c_uint32(0)      # last: a commit in flight lends the workspace until it is collected -- here: never, the context is destroyed under it
                N.check(lib.zkhip_kzg_commit_begin(h, N.ptr(srs.powers_of_tau_in_g1), None, N.ptr(srs.inf), C.c_size_t(1 << 10), N.ptr(sc), C.c_size_t(1 << 10),
                                                   C.c_int(1), C.byref(tk)), "commit_begin")
            N.check(lib.zkhip_ctx_destroy(h), "ctx_destroy")
        ctx = N.Context.get(0)
        ptrs = (C.c_void_p * 2)(*[x.data_ptr() for x in tabs])
        for k in range(n_sessions):
            st = C.c_void_p()
            N.check(lib.zkhip_sc_begin(ctx.handle, N.ptr(t), C.c_size_t(1 << 12), C.byref(st)), "sc_begin")
            if k % 3 == 0:
                N.check(lib.zkhip_sc_local_half_sums(st, N.ptr(torch.empty((2, 4), dtype=torch.int64, device="cuda"))), "half_sums")
            N.check(lib.zkhip_sc_abort(st), "sc_abort")
            ms = C.c_void_p()
            N.check(lib.zkhip_mc_begin(ctx.handle, ptrs, (C.c_uint32 * 1)(2), C.c_uint32(1), C.c_size_t(1 << 12), C.c_uint32(1), C.c_int(0), None, C.byref(ms)), "mc_begin")
            N.check(lib.zkhip_mc_abort(ms), "mc_abort")
            if k % 20 == 0:
                tk = C.c_uint32(0)
                N.check(lib.zkhip_kzg_commit_begin(ctx.handle, N.ptr(srs.powers_of_tau_in_g1), None, N.ptr(srs.inf), C.c_size_t(1 << 10), N.ptr(sc),
                                                   C.c_size_t(1 << 10), C.c_int(1), C.byref(tk)), "commit_begin")
                xy, inf = np.zeros(12, np.uint64), C.c_uint8(0)
                N.check(lib.zkhip_kzg_commit_end(ctx.handle, tk, p(xy), C.byref(inf)), "commit_end")

    cycle(20, 20)                                     # warm-up: grow-only buffers, kernel code objects, allocator pools reach their size
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(2000, 2000)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 >= free0 - (1 << 20), (free0, free1)     # nothing leaked (1 MiB of slack for allocator granularity)
    s = zk.Sumcheck(zk.Multilinear(ev))
    s.poly_sum()
    proof, ch = s.prove()
    ws, wrp, wch = ora.sumcheck_prove(ev)
    assert np.array_equal(proof.sum, ws) and np.array_equal(proof.univariate_poly, wrp) and np.array_equal(ch, wch)


# ---- a failing rank must not hang its peers (include/zkhip.h, csrc/shard_protocol.hpp) -----------------------------------------------
@pytest.mark.parametrize("world", [2, 4])
def test_failing_rank_reports_through_the_exchange_and_nobody_hangs(zk, ora, world):
    """Threads as ranks, real concurrent exchanges.  One rank fails in front of exchange i -- for EVERY i of the sumcheck and
    multi-composed protocols, two places of the GKR prover and the commit's one exchange -- or fails at begin (a busy workspace: no
    session at all).  That rank gets its own status, every other rank ZkhipPeerError, nobody waits for ever, and the next proof on the
    same communicators is bit-exact again."""
    import torch
    from zk_cryptography_amd import _native as N
    from zk_cryptography_amd import distributed as D
    xch = BarrierExchange(world)
    full = zk.Fr.random(1 << 21, 6150)                                   # shards of 2^20 / 2^19: the overlapped stage, 3 exchanges
    sc = zk.Sumcheck(zk.Multilinear(full))
    sc.poly_sum()
    want_sc, want_ch = sc.prove()
    tabs = [zk.Fr.random(1 << 15, 6160 + k) for k in range(4)]
    poly = [zk.ComposedMultilinear([zk.Multilinear(tabs[0]), zk.Multilinear(tabs[1])]),
            zk.ComposedMultilinear([zk.Multilinear(tabs[2]), zk.Multilinear(tabs[3])])]
    claimed = zk.MultiComposedSumcheckProver.calculate_poly_sum(poly)
    want_mc, want_mc_ch = zk.MultiComposedSumcheckProver.prove_partial(poly, claimed)
    want_mc = [p.monomials() for p in want_mc.round_polys]
    depth = 10
    gkr_in = zk.Fr.random(2 ** depth, 6170)
    circuit = zk.Circuit.random(depth)
    want_gkr = zk.GKRProtocol.prove(circuit, circuit.evaluation(gkr_in))
    srs = zk.TrustedSetup.setup(zk.Fr.random(10, 6180))
    scal = zk.Fr.random(1 << 10, 6181)
    want_c = zk.MultilinearKZG.commitment(zk.Multilinear(scal), srs)
    srs_xy, srs_inf = srs.powers_of_tau_in_g1.cpu(), srs.inf.cpu()
    torch.cuda.synchronize()
    seen = [[] for _ in range(world)]

    def body(rank):
        with torch.cuda.stream(torch.cuda.Stream()):
            ctx = N.Context.get(0)
            cb, err = xch.callback(ctx, rank)
            comm = D.Comm(ctx, world, rank, transport=cb)

            def cuda(a):
                return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()

            shard = cuda(full[rank::world])
            sh = [cuda(t[rank::world]) for t in tabs]
            my_circuit = zk.Circuit.random(depth)
            my_ev = my_circuit.evaluation(gkr_in)
            pts, pinf, psc = srs_xy[rank::world].contiguous().cuda(), srs_inf[rank::world].contiguous().cuda(), cuda(scal[rank::world])

            def prove_sc():
                e = D.ShardedSumcheck(D.HipSumcheckEngine(shard), world, comm=comm)
                s, rp, ch = e.prove()
                assert np.array_equal(s, want_sc.sum) and np.array_equal(rp, want_sc.univariate_poly) and np.array_equal(ch, want_ch)
                return e.exchanges

            def prove_mc():
                e = D.ShardedComposedSumcheck(D.HipComposedEngine([sh[:2], sh[2:]], world, multi=True, claimed_sum=claimed), world, comm=comm)
                rps, ch = e.prove()
                assert [zk.SparseUnivariatePolynomial(c_, p_).monomials() for c_, p_ in rps] == want_mc and np.array_equal(ch, want_mc_ch)
                return e.exchanges

            def prove_gkr():
                got = zk.GKRProtocol.prove_sharded(my_circuit, my_ev, world, rank, comm=comm)
                assert all(a.to_bytes() == b.to_bytes() for a, b in zip(got.sumcheck_proofs, want_gkr.sumcheck_proofs))
                return 2

            def commit():
                xy, inf = D.sharded_commit(pts, pinf, psc, comm)
                assert (not inf) and np.array_equal(xy, want_c.xy)
                return 1

            def expect_failure(run, failing, arm):
                """`arm()` on the failing rank makes its next run fail; every rank must come back with the right status"""
                if rank == failing:
                    arm()
                try:
                    run()
                    seen[rank].append("no status")
                except N.ZkhipPeerError:
                    seen[rank].append("peer" if rank != failing else "WRONG: the failed rank saw ERR_PEER")
                except N.ZkhipError as e:
                    seen[rank].append("own %d" % e.status if rank == failing else "WRONG: status %r on a healthy rank" % e.status)

            try:
                for run in (prove_sc, prove_mc, prove_gkr, commit):
                    n_ex = run()                                             # healthy first: how many exchanges there are to fail in front of
                    for idx in range(n_ex):
                        expect_failure(run, idx % world, lambda idx=idx: comm.inject_failure(idx, N.ERR_NOMEM))
                    run()                                                    # and healthy again, on the same communicators
                # a failure at BEGIN: the rank's workspace is lent to another session, zkhip_mc_begin gives ZKHIP_ERR_BUSY -- no session,
                # the whole exchange schedule is walked from the shapes alone
                hold = []

                def lend_workspace():
                    st = C.c_void_p()
                    N.check(N.lib().zkhip_sc_begin(ctx.handle, N.ptr(shard), C.c_size_t(shard.shape[0]), C.byref(st)), "sc_begin")
                    hold.append(st)

                def mc_one_call():
                    ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in sh])
                    n_rounds = 15
                    lens, rp, ch = np.zeros(n_rounds, np.uint32), np.zeros((n_rounds, 7, 2, 4), np.uint64), np.zeros((n_rounds, 4), np.uint64)
                    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
                    comm.check(N.lib().zkhip_multi_composed_prove_sharded(comm.handle, ptrs, (C.c_uint32 * 2)(2, 2), C.c_uint32(2), C.c_size_t(sh[0].shape[0]),
                                                                          p(np.ascontiguousarray(claimed)), C.c_int(-1), p(lens), p(rp), p(ch), None), "mc_sharded")
                    assert np.array_equal(ch, want_mc_ch)
                expect_failure(mc_one_call, world - 1, lend_workspace)
                for st in hold:
                    N.check(N.lib().zkhip_sc_abort(st), "sc_abort")
                mc_one_call()
            finally:
                if err:
                    raise err[0]
                comm.close()

    _run_threads(world, body)
    # every injected case: exactly one rank with its own status (-5; -6 for the busy begin), everybody else "peer"
    n_cases = len(seen[0])
    assert n_cases >= 3 + 3 + 2 + 1 + 1 and all(len(s) == n_cases for s in seen)   # every exchange of the sumcheck (3) and multi-composed protocols, 2 in GKR, commit, busy begin
    for k in range(n_cases):
        col = [seen[r][k] for r in range(world)]
        assert col.count("peer") == world - 1, (k, col)
        assert [c for c in col if c != "peer"][0] in ("own -5", "own -6"), (k, col)

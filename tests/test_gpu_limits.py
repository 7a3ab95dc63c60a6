"""Every documented hard limit of the C ABI returns its documented status (never undefined behaviour), and the shapes where
this implementation and the reference both have a defined result agree (the empty table of partial_evaluation)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import zk_cryptography_amd as z
    return z


def F(zk, v):
    return zk.Fr.from_ints(v)


@pytest.mark.parametrize("n,k", [(8, 3), (16, 4), (16, 7), (64, 6), (64, 31)])
def test_partial_evaluation_empty_table(zk, n, k):
    """2^k >= n with k < n/2: pick_pairs_with_random_index (polynomial/src/utils.rs:26-53) yields no pair, so the reference
    returns an EMPTY table with n_vars - 1 (evaluation_form.rs:137-140)."""
    p = zk.Multilinear(F(zk, list(range(n))))
    out = p.partial_evaluation(F(zk, [5])[0], k)
    assert len(out) == 0 and out.n_vars == p.n_vars - 1
    from zk_cryptography_amd import _native as N
    N.lib().zkhip_mle_partial_evaluation_len.restype = C.c_size_t
    assert N.lib().zkhip_mle_partial_evaluation_len(C.c_size_t(n), C.c_uint32(k)) == 0
    assert N.lib().zkhip_mle_partial_evaluation_len(C.c_size_t(n), C.c_uint32(0)) == n // 2
    with pytest.raises(AssertionError):                      # k >= n/2 is the reference's assert (utils.rs:31-34)
        p.partial_evaluation(F(zk, [5])[0], n // 2)
    # partial_evaluations: fine as the LAST fold, the reference's panic when another fold follows
    with pytest.raises(AssertionError):
        p.partial_evaluations(F(zk, [5, 6]), [k, 0])


def test_composed_limits(zk):
    from zk_cryptography_amd import _native as N
    tabs = [F(zk, [1, 2, 3, 4]) for _ in range(6)]
    with pytest.raises(N.ZkhipError):                        # K <= 5 tables per product term (zkhip.h)
        zk.ComposedSumcheck(zk.ComposedMultilinear(tabs)).prove()
    with pytest.raises(N.ZkhipError):
        zk.ComposedSumcheck.calculate_poly_sum(zk.ComposedMultilinear(tabs))
    terms = [zk.ComposedMultilinear([F(zk, [1, 2, 3, 4])]) for _ in range(5)]
    s = np.zeros(4, dtype=np.uint64)
    with pytest.raises(N.ZkhipError):                        # at most 4 terms
        zk.MultiComposedSumcheckProver.prove_partial(terms, s)
    # 4 terms of degree 5 = 24 sums per record: beyond the 16 a round record holds -> refused, not truncated
    big = [zk.ComposedMultilinear([F(zk, [1, 2, 3, 4]) for _ in range(5)]) for _ in range(4)]
    with pytest.raises(N.ZkhipError):
        zk.MultiComposedSumcheckProver.prove_partial(big, s)
    # the element-wise vectors have no such cap
    assert len(zk.ComposedMultilinear(tabs).element_wise_product()) == 4


def test_gkr_depth_limit(zk):
    """2 * n_layers <= 48 sumcheck rounds per layer proof: depth 25 is refused with the shape status before any work (depth 21 .. 24:
    tests/test_gpu_baseline_sizes.py::test_gkr_beyond_depth_20)."""
    from zk_cryptography_amd import _native as N
    n_layers = 25
    n_gates = (C.c_size_t * n_layers)(*[1] * n_layers)
    gt = (C.c_uint8 * n_layers)()
    z = (C.c_uint32 * n_layers)()
    ctx = N.Context.get()
    cir = C.c_void_p()
    st = N.lib().zkhip_circuit_create(ctx.handle, C.c_uint32(n_layers), n_gates, gt, z, z, C.byref(cir))
    assert st == N.ERR_SHAPE


def test_srs_table_size_limit(zk):
    """n * 13 table entries must stay below 2^31: refused with the shape status (no allocation happens for the check)."""
    from zk_cryptography_amd import _native as N
    ctx = N.Context.get()
    n = (1 << 31) // 13 + 1
    dummy = C.c_void_p(16)          # never dereferenced: the size check comes first
    st = N.lib().zkhip_srs_precompute(ctx.handle, dummy, None, C.c_size_t(n), dummy)
    assert st == N.ERR_SHAPE
    out = np.zeros(12, dtype=np.uint64)
    inf = C.c_uint8(0)
    st = N.lib().zkhip_kzg_commit_table(ctx.handle, dummy, None, C.c_size_t(n), dummy, C.c_size_t(n), C.c_int(1),
                                        out.ctypes.data_as(C.c_void_p), C.byref(inf))
    assert st == N.ERR_SHAPE


def test_elementwise_shorter_rhs_is_the_index_panic(zk):
    a, b = zk.Multilinear(F(zk, [1, 2, 3, 4])), zk.Multilinear(F(zk, [1, 2]))
    with pytest.raises(IndexError):                          # rhs.evaluations[i] out of bounds (evaluation_form.rs:185)
        a + b
    with pytest.raises(IndexError):
        a - b
    assert zk.Fr.to_ints((b + a).to_numpy()) == [2, 4]       # a longer rhs is read up to len(self)


def test_workspace_is_busy_while_a_session_is_live(zk, ora):
    """A split-phase session borrows the context's workspace: other entry points that need it return ZKHIP_ERR_BUSY until the
    session is finished or aborted (include/zkhip.h)."""
    import torch
    from zk_cryptography_amd import _native as N
    from zk_cryptography_amd import distributed as D
    t = torch.from_numpy(ora.random_fr(1 << 12, 3).view(np.int64)).cuda()
    eng = D.HipSumcheckEngine(t)
    sc = zk.Sumcheck(zk.Multilinear(ora.random_fr(1 << 12, 4)))
    with pytest.raises(N.ZkhipError, match="split-phase"):
        sc.prove()
    eng.abort()
    sc.poly_sum()
    proof, ch = sc.prove()
    s, rp, och = ora.sumcheck_prove(sc.poly.to_numpy())
    assert np.array_equal(proof.univariate_poly, rp) and np.array_equal(ch, och)
    # a dropped engine releases its state too
    eng = D.HipSumcheckEngine(t)
    del eng
    sc.prove()


def test_tables_carry_a_header_that_is_checked_against_the_geometry(zk, ora):
    """A shifted-SRS table / the level tables begin with a 128-byte header (magic, kind, points, window widths): handed to an entry point
    that would address them with another geometry they are refused (ZKHIP_ERR_ARG) instead of giving a wrong commitment."""
    import torch
    from zk_cryptography_amd import _native as N
    lib = N.lib()
    lib.zkhip_srs_table_bytes.restype = C.c_size_t
    lib.zkhip_srs_level_tables_bytes.restype = C.c_size_t
    junk = torch.zeros(lib.zkhip_srs_table_bytes(C.c_size_t(1 << 10)), dtype=torch.uint8, device="cuda")
    N.check(lib.zkhip_table_release(N.Context.get(0).handle, N.ptr(junk)), "table_release")   # (whatever an earlier test's table left behind at this address)
    srs10 = zk.TrustedSetup.setup(ora.random_fr(10, 4501)).precompute().precompute_open()
    srs9 = zk.TrustedSetup.setup(ora.random_fr(9, 4502)).precompute()
    ctx = N.Context.get(0)
    sc = torch.from_numpy(ora.random_fr(1 << 10, 4503).view(np.int64)).cuda()
    xy, inf = np.zeros(12, np.uint64), C.c_uint8(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731

    def commit(table, srs, n_points, n_scalars):
        return lib.zkhip_kzg_commit_table(ctx.handle, N.ptr(table), N.ptr(srs.inf), C.c_size_t(n_points), N.ptr(sc), C.c_size_t(n_scalars), C.c_int(0), p(xy), C.byref(inf))
    assert commit(srs10._table, srs10, 1 << 10, 1 << 10) == N.ZKHIP_OK
    want = zk.MultilinearKZG.commitment(zk.Multilinear(sc), zk.TrustedSetup(srs10.powers_of_tau_in_g1, srs10.inf))      # no table: the bucket pipeline on the points
    assert np.array_equal(xy, want.xy)
    assert commit(srs10._table, srs10, 1 << 9, 1 << 9) == N.ERR_ARG              # a table built for 2^10 points addressed as one of 2^9
    assert commit(srs9._table, srs10, 1 << 10, 1 << 10) == N.ERR_ARG             # ... and the other way round (the header is read, nothing behind it)
    assert commit(srs10._level_tables, srs10, 1 << 10, 1 << 10) == N.ERR_ARG     # level tables are not a shifted-SRS table
    assert commit(junk, srs10, 1 << 10, 1 << 10) == N.ERR_ARG                    # no table at all
    tk = C.c_uint32(0)
    assert lib.zkhip_kzg_commit_begin(ctx.handle, None, N.ptr(srs9._table), N.ptr(srs10.inf), C.c_size_t(1 << 10), N.ptr(sc), C.c_size_t(1 << 10), C.c_int(0),
                                      C.byref(tk)) == N.ERR_ARG
    # openings: the shifted-SRS table where the level tables belong
    ev = torch.from_numpy(ora.random_fr(1 << 10, 4504).view(np.int64)).cuda()
    z = ora.random_fr(10, 4505)
    fxy, finf = srs10.folded()
    h_ev, pxy, pinf = np.zeros(4, np.uint64), np.zeros((10, 12), np.uint64), np.zeros(10, np.uint8)

    def open_with(tables):
        return lib.zkhip_kzg_open_tables(ctx.handle, N.ptr(ev), C.c_size_t(1 << 10), p(z), C.c_size_t(10), N.ptr(srs10.powers_of_tau_in_g1), N.ptr(srs10.inf),
                                         C.c_size_t(1 << 10), N.ptr(fxy), N.ptr(finf), N.ptr(tables), p(h_ev), p(pxy), p(pinf))
    assert open_with(srs10._level_tables) == N.ZKHIP_OK
    assert np.array_equal(h_ev, ora.mle_evaluation(ev.cpu().numpy().view(np.uint64), z))
    assert open_with(srs10._table) == N.ERR_ARG
    # a table REBUILT at the same address for another size is checked again, not remembered
    N.check(lib.zkhip_srs_precompute(ctx.handle, N.ptr(srs9.powers_of_tau_in_g1), N.ptr(srs9.inf), C.c_size_t(1 << 9), N.ptr(srs10._table)), "precompute")
    assert commit(srs10._table, srs10, 1 << 10, 1 << 10) == N.ERR_ARG
    assert commit(srs10._table, srs9, 1 << 9, 1 << 9) == N.ZKHIP_OK
    # a table that is DROPPED releases its address: the allocator hands it to a buffer that is no table, and that buffer is read and refused
    srs8 = zk.TrustedSetup.setup(ora.random_fr(8, 4506)).precompute()
    sc8 = torch.from_numpy(ora.random_fr(1 << 8, 4507).view(np.int64)).cuda()
    commit8 = lambda table: lib.zkhip_kzg_commit_table(ctx.handle, N.ptr(table), N.ptr(srs8.inf), C.c_size_t(1 << 8), N.ptr(sc8), C.c_size_t(1 << 8), C.c_int(0), p(xy), C.byref(inf))   # noqa: E731
    assert commit8(srs8._table) == N.ZKHIP_OK
    addr, nbytes = srs8._table.data_ptr(), srs8._table.numel()
    inf8 = srs8.inf
    srs8.invalidate()                                                             # the table goes (zkhip_table_release), its memory back to the caching allocator
    again = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    if again.data_ptr() == addr:                                                  # (the allocator is free to choose: checked when it does reuse the block)
        assert lib.zkhip_kzg_commit_table(ctx.handle, N.ptr(again), N.ptr(inf8), C.c_size_t(1 << 8), N.ptr(sc8), C.c_size_t(1 << 8), C.c_int(0), p(xy), C.byref(inf)) == N.ERR_ARG

"""GPU parity: sumcheck::Sumcheck::prove on the HIP path (device transcript) vs the CPU oracle,
bit-exact on sum, round polynomials and challenges.  Names follow sumcheck/src/sumcheck.rs tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import zk_cryptography_amd as z
    return z


def _prove(zk, evals):
    sc = zk.Sumcheck(zk.Multilinear(evals))
    sc.poly_sum()
    proof, ch = sc.prove()
    return sc, proof, ch


def test_sum_calculation(zk):
    sc = zk.Sumcheck(zk.Multilinear(zk.Fr.from_ints([0, 0, 0, 2, 2, 2, 2, 4])))
    sc.poly_sum()
    assert zk.Fr.to_ints(sc.sum) == [12]


@pytest.mark.parametrize("vals", [
    [0, 0, 2, 7, 3, 3, 6, 11],
    [0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0],
    [1, 3, 5, 7, 2, 4, 6, 8, 3, 5, 7, 9, 4, 6, 8, 10],
])
def test_sum_check_proof(zk, ora, vals):
    ev = zk.Fr.from_ints(vals)
    sc, proof, ch = _prove(zk, ev)
    s, rp, och = ora.sumcheck_prove(ev)
    assert np.array_equal(proof.sum, s) and np.array_equal(proof.univariate_poly, rp) and np.array_equal(ch, och)
    assert ora.sumcheck_verify(ev, proof.sum, proof.univariate_poly)      # restated verifier accepts the GPU proof


@pytest.mark.parametrize("log_n", [0, 1, 2, 9, 10, 11, 12, 14, 17])
def test_prove_matches_oracle_random(zk, ora, log_n):
    ev = ora.random_fr(1 << log_n, 4000 + log_n)
    sc, proof, ch = _prove(zk, ev)
    s, rp, och = ora.sumcheck_prove(ev)
    assert np.array_equal(proof.sum, s)
    assert np.array_equal(proof.univariate_poly, rp)
    assert np.array_equal(ch, och)


def test_prove_without_poly_sum_absorbs_default_sum(zk, ora):
    # the reference's prove(&self) absorbs self.sum = Default (zero) if poly_sum() was never called
    ev = ora.random_fr(1 << 12, 17)
    sc = zk.Sumcheck(zk.Multilinear(ev))
    proof, ch = sc.prove()
    assert zk.Fr.to_ints(proof.sum) == [0]
    import hashlib
    R = zk.Fr.MODULUS
    hs = ora.mle_half_sums(ev)
    d = hashlib.sha256(bytes(32) + ora.fr_to_bytes_be(hs[0]) + ora.fr_to_bytes_be(hs[1])).digest()
    assert zk.Fr.to_ints(ch[0]) == [int.from_bytes(d, "big") % R]
    assert np.array_equal(proof.univariate_poly[0], hs)


def test_prove_2_24_self_consistency(zk, ora):
    """BASELINE config 2 size.  The oracle prover would take ~10 s here; instead check the proof
    with the verifier's own equations: p_i(0)+p_i(1) = claim_i, claim_{i+1} = p_i(r_i) (linear
    interpolation), final claim = f(r) via the GPU evaluation, and the transcript via hashlib."""
    import hashlib
    import torch
    n = 1 << 24
    g = torch.Generator(device="cuda").manual_seed(99)
    t = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    poly = zk.Multilinear(t)
    sc = zk.Sumcheck(poly)
    sc.poly_sum()
    proof, ch = sc.prove()
    R = zk.Fr.MODULUS
    claim = zk.Fr.to_ints(proof.sum)[0]
    h = hashlib.sha256(claim.to_bytes(32, "big"))
    for i in range(24):
        p0, p1 = zk.Fr.to_ints(proof.univariate_poly[i])
        assert (p0 + p1) % R == claim, "round %d" % i
        h.update(p0.to_bytes(32, "big") + p1.to_bytes(32, "big"))
        d = h.digest()
        r = int.from_bytes(d, "big") % R
        assert zk.Fr.to_ints(ch[i]) == [r], "challenge %d" % i
        h = hashlib.sha256(d)
        claim = (p0 + r * (p1 - p0)) % R
    assert zk.Fr.to_ints(poly.evaluation(ch)) == [claim]


@pytest.mark.parametrize("depth", [2, 4, 8])
def test_proofs_in_flight_match_synchronous_proofs(zk, ora, depth):
    """zkhip_sumcheck_prove_begin / _end: up to eight proofs in flight on lanes of their own (streams, workspace, scratch),
    different tables and sizes -- overlapped and generic plans side by side -- each equal to the oracle's."""
    from zk_cryptography_amd import _native as N
    tables = [ora.random_fr(1 << log_n, 5100 + log_n) for log_n in (20, 12, 21, 3, 19, 20, 22, 9)]
    want = [ora.sumcheck_prove(t) for t in tables]
    got, pending = [], []
    for rep in range(2):                       # twice: the second pass reuses every lane
        for t in tables:
            sc = zk.Sumcheck(zk.Multilinear(t))
            sc.poly_sum()
            pending.append(sc.prove_begin())
            if len(pending) == depth:
                got.append(pending.pop(0).wait())
        got += [h.wait() for h in pending]
        pending = []
    for (proof, ch), (s, rp, och) in zip(got, want + want):
        assert np.array_equal(proof.sum, s) and np.array_equal(proof.univariate_poly, rp) and np.array_equal(ch, och)
    # a ninth proof in flight and a synchronous prove are refused while eight are pending
    scs = [zk.Sumcheck(zk.Multilinear(t)) for t in tables + tables[:1]]
    held = [sc.prove_begin() for sc in scs[:8]]
    with pytest.raises(N.ZkhipError):
        scs[8].prove_begin()
    with pytest.raises(N.ZkhipError):
        scs[8].prove()
    held[0].wait()
    del held
    scs[8].prove()                             # every ticket is free again


def test_proofs_in_flight_without_poly_sum_and_same_table(zk, ora):
    """In flight without poly_sum() (the prover derives fine and coarse sums itself, on its lane) and two proofs of ONE table
    begun back to back (same block sums read by both)."""
    t = ora.random_fr(1 << 20, 777)
    s, rp, och = ora.sumcheck_prove(t)
    poly = zk.Multilinear(t)
    a = zk.Sumcheck(poly)
    a.poly_sum()
    b = zk.Sumcheck(poly)
    b._block_sums, b._log_blocks, b._sum_deferred = a._block_sums, a._log_blocks, True    # the same device block sums, no second pass
    ha, hb = a.prove_begin(), b.prove_begin()
    for proof, ch in (ha.wait(), hb.wait()):
        assert np.array_equal(proof.sum, s) and np.array_equal(proof.univariate_poly, rp) and np.array_equal(ch, och)
    c = zk.Sumcheck(zk.Multilinear(t))
    c.sum = s                                    # claimed sum given by the caller, no block sums at all
    proof, ch = c.prove_begin().wait()
    assert np.array_equal(proof.univariate_poly, rp) and np.array_equal(ch, och)


def test_proofs_in_flight_at_2_24_keep_their_folds_on_the_callers_stream(zk):
    """From the third proof in flight on, the big fold of an overlapped-plan proof (2^24 entries) is held back and enqueued on the caller's
    stream behind the next tables' sums passes (zkhip_ctx::deferred).  Every way the held-back half can be flushed -- by the next begin, by
    wait() in order, out of order and of a proof that still holds it, by a synchronize of the context, by dropping the ticket -- must
    deliver the synchronous prove()'s proof, bit for bit."""
    import torch
    from zk_cryptography_amd import _native as N
    g = torch.Generator(device="cuda")
    g.manual_seed(2424)
    polys = [zk.Multilinear(torch.randint(0, 2 ** 62, (1 << 24, 4), dtype=torch.int64, device="cuda", generator=g)) for _ in range(5)]
    want = []
    for pl in polys:
        sc = zk.Sumcheck(pl)
        sc.poly_sum()
        want.append(sc.prove())

    def same(got, j):
        (proof, ch), (wproof, wch) = got, want[j]
        return np.array_equal(proof.sum, wproof.sum) and np.array_equal(proof.univariate_poly, wproof.univariate_poly) and np.array_equal(ch, wch)

    def begin(j):
        sc = zk.Sumcheck(polys[j])
        sc.poly_sum()
        return sc.prove_begin()

    for depth in (3, 4, 5, 6, 8):                                  # the pipeline of bench.py's `pipelined` leg, filled and drained
        pend = []
        for i in range(2 * depth + 3):
            pend.append((i % 5, begin(i % 5)))
            if len(pend) == depth:
                j, h = pend.pop(0)
                assert same(h.wait(), j), (depth, i)
        for j, h in pend:
            assert same(h.wait(), j), depth
    hs = [begin(j) for j in range(5)]                             # collected youngest first: the first wait() flushes every half held back
    for j in reversed(range(5)):
        assert same(hs[j].wait(), j)
    hs = [begin(j) for j in range(4)]
    assert same(hs[3].wait(), 3) and same(hs[0].wait(), 0)         # ... and from the middle
    assert same(hs[2].wait(), 2) and same(hs[1].wait(), 1)
    hs = [begin(j) for j in range(5)]
    N.Context.get().synchronize()                                  # drains the caller's stream: nothing may stay behind
    torch.cuda.synchronize()
    for j in range(5):
        assert same(hs[j].wait(), j)
    hs = [zk.Sumcheck(polys[j]) for j in range(5)]                # without poly_sum(): the sums pass is the prover's own, on the caller's stream from the third proof on
    for sc in hs:
        sc._sum_deferred = True                                   # (absorb the TRUE sum, as after poly_sum(): no claimed sum, no block sums handed in)
    hs = [sc.prove_begin() for sc in hs]
    for j in range(5):
        assert same(hs[j].wait(), j)
    hs = [begin(j) for j in range(4)]                             # the caller moves to another stream with halves held back: they go onto the old one first
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hs.append(begin(4))                                       # (Context.get() follows torch's current stream: zkhip_ctx_set_stream)
        for j in (4, 0, 2):
            assert same(hs[j].wait(), j)
    assert same(hs[1].wait(), 1) and same(hs[3].wait(), 3)        # ... and back on the default stream
    torch.cuda.synchronize()
    hs = [begin(j) for j in range(5)]                             # tickets dropped uncollected: waited out, lanes and slots free again
    del hs
    sc = zk.Sumcheck(polys[1])
    sc.poly_sum()
    assert same(sc.prove(), 1)


def test_proofs_in_flight_with_the_overlapped_plan_at_small_sizes():
    """The in-flight tests above once more in a process where the overlapped plan starts at 2^19 (ZKHIP_OVERLAP_MIN_LOG) instead of 2^24:
    overlapped, stage and serial plans side by side in one pipeline, folds held back on the caller's stream, against the oracle."""
    import os, subprocess, sys
    env = dict(os.environ, ZKHIP_OVERLAP_MIN_LOG="19")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k",
                          "test_proofs_in_flight_match_synchronous_proofs or test_proofs_in_flight_without_poly_sum_and_same_table or test_prove_matches_oracle_random"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0, out.stdout.decode()[-3000:]


@pytest.mark.parametrize("log_n", [21, 24])
def test_poly_sum_with_the_total_deferred(zk, ora, log_n):
    """Tables of 2^24 entries (the overlapped plan; the stage plan below that since round 5: tools/step_sizes.py): poly_sum() leaves the
    total to prove()'s own sum tree (zkhip_mle_block_sums_deferred).  The proof absorbs the true sum either way, `sum` read before or
    after the proof is the true sum, and a sum the caller overrides is absorbed as given -- at 2^21 (total computed by poly_sum) too."""
    evals = ora.random_fr(1 << log_n, 7100 + log_n)
    ws, wrp, wch = ora.sumcheck_prove(evals)
    poly = zk.Multilinear(evals)
    sc = zk.Sumcheck(poly)
    sc.poly_sum()
    assert sc._sum_deferred == (log_n == 24)
    proof, ch = sc.prove()                                   # (2^24: the total never computed by poly_sum)
    assert np.array_equal(proof.sum, ws) and np.array_equal(proof.univariate_poly, wrp) and np.array_equal(ch, wch)
    assert np.array_equal(sc.sum, ws)                        # ... and delivered on demand
    sc2 = zk.Sumcheck(poly)
    sc2.poly_sum()
    assert np.array_equal(sc2.sum, ws)                       # read first, then prove
    proof2, ch2 = sc2.prove()
    assert np.array_equal(proof2.univariate_poly, wrp) and np.array_equal(ch2, wch)
    sc3 = zk.Sumcheck(poly)
    sc3.poly_sum()
    sc3.sum = zk.Fr.from_int(5)                              # an overridden sum is absorbed as it is
    proof3, ch3 = sc3.prove()
    assert zk.Fr.to_ints(proof3.sum) == [5] and not np.array_equal(ch3, wch)
    assert np.array_equal(proof3.univariate_poly[0], wrp[0])

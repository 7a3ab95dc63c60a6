"""Randomised parity stress against the CPU oracle (test infrastructure): random shapes and sizes for a given time budget.
Usage: python tools/stress_parity.py [seconds] [seed].  Prints one line per failure and a summary; exit code = #failures."""
import sys, os, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import zk_cryptography_amd as zk
from oracle import oracle as ora

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails, counts = 0, {}
BIG = 4 if os.environ.get("STRESS_BIG") else 0   # STRESS_BIG=1: table sizes up to 16x larger (fewer, slower cases)
dev = lambda t: t.cpu().numpy().view(np.uint64)


def case_sumcheck():
    log_n = rng.randint(0, 19 + BIG)
    t = ora.random_fr(1 << log_n, rng.randrange(1 << 30))
    sc = zk.Sumcheck(zk.Multilinear(t))
    if rng.random() < 0.8: sc.poly_sum()
    proof, ch = sc.prove()
    if sc._sum_dev is None and not np.any(sc.sum):
        return "skip"   # prove() without poly_sum absorbs a zero sum; the oracle call below assumes the true sum
    ws, wrp, wch = ora.sumcheck_prove(t)
    return np.array_equal(proof.sum, ws) and np.array_equal(proof.univariate_poly, wrp) and np.array_equal(ch, wch), ("sumcheck", log_n)


def case_fold_eval():
    log_n = rng.randint(1, 18 + BIG)
    t = ora.random_fr(1 << log_n, rng.randrange(1 << 30))
    k = rng.randrange(log_n)
    r = ora.random_fr(1, rng.randrange(1 << 30))[0]
    p = zk.Multilinear(t)
    ok = np.array_equal(dev(p.partial_evaluation(r, k).evaluations), ora.mle_partial_evaluation(t, r, k))
    pts = ora.random_fr(log_n, rng.randrange(1 << 30))
    ok = ok and np.array_equal(p.evaluation(pts), ora.mle_evaluation(t, pts))
    return ok, ("fold/eval", log_n, k)


def case_composed():
    k, log_n = rng.randint(1, 5), rng.randint(1, 14 + BIG)
    t = np.stack([ora.random_fr(1 << log_n, rng.randrange(1 << 30)) for _ in range(k)])
    proof, ch = zk.ComposedSumcheck(zk.ComposedMultilinear(list(t))).prove()
    rp, och = ora.composed_prove(t)
    return np.array_equal(proof.round_polys, rp) and np.array_equal(ch, och), ("composed", k, log_n)


def case_multi():
    n_terms = rng.randint(1, 4)
    sizes = [rng.randint(1, 5) for _ in range(n_terms)]
    while sum(s + 1 for s in sizes) > 16: sizes.pop()
    log_n = rng.randint(1, 12 + BIG)
    flat = np.stack([ora.random_fr(1 << log_n, rng.randrange(1 << 30)) for _ in range(sum(sizes))])
    if rng.random() < 0.3: flat[rng.randrange(len(flat))][:] = 0          # a zero table: zero coefficients dropped per term
    terms, q = [], 0
    for k in sizes:
        terms.append(zk.ComposedMultilinear([zk.Multilinear(flat[q + i]) for i in range(k)])); q += k
    s = zk.MultiComposedSumcheckProver.calculate_poly_sum(terms)
    partial = rng.random() < 0.5
    fn = zk.MultiComposedSumcheckProver.prove_partial if partial else zk.MultiComposedSumcheckProver.prove
    proof, ch = fn(terms, s)
    orps, och = ora.multi_composed_prove(flat, sizes, s, partial=partial)
    return (np.array_equal(s, ora.multi_composed_sum(flat, sizes)) and proof.to_bytes() == ora.multi_composed_proof_bytes(orps)
            and np.array_equal(ch, och)), ("multi", sizes, log_n, partial)


def case_multi_k2():
    """claims whose terms are all products of TWO tables: the provers run one round ahead of the transcript (csrc/composed_pipe.hpp) --
    pipelined launches above the LDS tail, the pipelined tail below; degenerate tables make coefficients vanish (dropped per term)"""
    n_terms = rng.randint(1, 4)
    log_n = rng.randint(1, 15 + min(BIG, 2))
    n = 1 << log_n
    flat = np.stack([ora.random_fr(n, rng.randrange(1 << 30)) for _ in range(2 * n_terms)])
    for q in range(2 * n_terms):
        kind = rng.random()
        if kind < 0.12: flat[q][:] = 0                                                # a zero table: every coefficient of its term vanishes
        elif kind < 0.24: flat[q][:] = flat[q][0]                                     # a constant table: the term's x^2 coefficient vanishes
        elif kind < 0.32: flat[q] = zk.Fr.from_ints([(i * 3 + 1) % 7 for i in range(n)])    # small values
        elif kind < 0.38 and q > 0: flat[q] = flat[q - 1]                             # the same table twice (GKR: V in both terms)
    sizes = [2] * n_terms
    terms = [zk.ComposedMultilinear([zk.Multilinear(flat[2 * p]), zk.Multilinear(flat[2 * p + 1])]) for p in range(n_terms)]
    if n_terms == 1 and rng.random() < 0.5:          # ComposedSumcheck (raw evaluations absorbed)
        proof, ch = zk.ComposedSumcheck(terms[0]).prove()
        rp, och = ora.composed_prove(flat)
        return np.array_equal(proof.round_polys, rp) and np.array_equal(ch, och), ("composed_k2", log_n)
    s = zk.MultiComposedSumcheckProver.calculate_poly_sum(terms)
    partial = rng.random() < 0.7
    fn = zk.MultiComposedSumcheckProver.prove_partial if partial else zk.MultiComposedSumcheckProver.prove
    proof, ch = fn(terms, s)
    orps, och = ora.multi_composed_prove(flat, sizes, s, partial=partial)
    return (proof.to_bytes() == ora.multi_composed_proof_bytes(orps) and np.array_equal(ch, och)), ("multi_k2", n_terms, log_n, partial)


_srs = {}
def case_commit():
    nv = rng.randint(1, 10)
    if nv not in _srs:
        tau = ora.random_fr(nv, 1000 + nv)
        _srs[nv] = (zk.TrustedSetup.setup(tau), ora.kzg_multilinear_srs_g1(tau))
    srs, osrs = _srs[nv]
    kind = rng.choice(["uniform", "small", "sparse", "same"])
    n = 1 << nv
    if kind == "uniform": v = ora.random_fr(n, rng.randrange(1 << 30))
    elif kind == "small": v = zk.Fr.from_ints([rng.randrange(0, 300) for _ in range(n)])
    elif kind == "sparse": v = zk.Fr.from_ints([rng.randrange(1 << 200) if rng.random() < 0.1 else 0 for _ in range(n)])
    else: v = zk.Fr.from_ints([zk.Fr.MODULUS - 1] * n)
    if rng.random() < 0.5:          # half of them against the shifted-SRS table
        srs.precompute()
    elif srs.table is not None:
        srs.invalidate()
    com = zk.MultilinearKZG.commitment(zk.Multilinear(v), srs)
    want = ora.g1_to_affine(ora.kzg_commitment(v, osrs, True))
    return (bool(want[12]) == com.infinity) and (com.infinity or np.array_equal(com.xy, want[:12])), ("commit", nv, kind, srs.table is not None)


def case_ntt():
    log_n = rng.randint(0, 15 + BIG)
    x = ora.random_fr(1 << log_n, rng.randrange(1 << 30))
    d = zk.Domain(1 << log_n)
    ok = np.array_equal(dev(d.fft(x)), ora.domain_fft(x, 1 << log_n)) and np.array_equal(dev(d.ifft(x)), ora.domain_ifft(x, 1 << log_n))
    na, nb = rng.randint(1, 3000), rng.randint(1, 3000)
    a, b = ora.random_fr(na, rng.randrange(1 << 30)), ora.random_fr(nb, rng.randrange(1 << 30))
    got = dev(zk.UnivariateEval.multiply(zk.DenseUnivariatePolynomial(a), zk.DenseUnivariatePolynomial(b)).coefficients)
    return ok and np.array_equal(got, ora.univariate_multiply(a, b)), ("ntt", log_n, na, nb)


def case_gkr():
    from gkr_cases import A, M
    depth = rng.randint(1, 6 + BIG // 2)
    layers = []
    for li in range(depth):
        n_in = 2 ** (li + 1)
        layers.append([(rng.choice((A, M)), rng.randrange(n_in), rng.randrange(n_in)) for _ in range(2 ** li)])
    if depth >= 1: layers[0] = layers[0][:1]
    inp = ora.random_fr(2 ** depth, rng.randrange(1 << 30))
    circuit = zk.Circuit.from_tuples(layers)
    ev = circuit.evaluation(inp)
    proof = zk.GKRProtocol.prove(circuit, ev)
    want = ora.gkr_prove(layers, ora.circuit_evaluation(layers, inp))
    ok = True
    for k, sp in enumerate(proof.sumcheck_proofs):
        w_sum, w_rps, w_wb, w_wc = want.layer(k)
        ok = ok and np.array_equal(sp.sum, w_sum) and sp.to_bytes() == ora.multi_composed_proof_bytes(w_rps)
        ok = ok and np.array_equal(proof.wb_s[k], w_wb) and np.array_equal(proof.wc_s[k], w_wc)
    return ok, ("gkr", depth)


def case_gkr_batch():
    """zkhip_gkr_prove_batch: several inputs of one random circuit in one call (lanes, replayed launch chains) against the oracle, proof by proof;
    the circuit changes from case to case, so the lanes' recorded graphs are dropped and re-recorded all the time"""
    from gkr_cases import A, M
    depth = rng.randint(1, 6 + BIG // 2)
    layers = []
    for li in range(depth):
        n_in = 2 ** (li + 1)
        layers.append([(rng.choice((A, M)), rng.randrange(n_in), rng.randrange(n_in)) for _ in range(2 ** li)])
    layers[0] = layers[0][:1]
    circuit = zk.Circuit.from_tuples(layers)
    inputs = [ora.random_fr(2 ** depth, rng.randrange(1 << 30)) for _ in range(rng.randint(1, 20))]
    ok = True
    for rep in range(rng.randint(1, 2)):
        proofs = zk.GKRProtocol.prove_batch(circuit, [circuit.evaluation(i) for i in inputs], max_lanes=rng.choice([0, 1, 3, 8, 12]))
        for inp, proof in zip(inputs, proofs):
            want = ora.gkr_prove(layers, ora.circuit_evaluation(layers, inp))
            for k, sp in enumerate(proof.sumcheck_proofs):
                w_sum, w_rps, w_wb, w_wc = want.layer(k)
                ok = ok and np.array_equal(sp.sum, w_sum) and sp.to_bytes() == ora.multi_composed_proof_bytes(w_rps)
                ok = ok and np.array_equal(proof.wb_s[k], w_wb) and np.array_equal(proof.wc_s[k], w_wc)
    return ok, ("gkr_batch", depth, len(inputs))


def case_open():
    nv = rng.randint(2, 5)
    if nv not in _srs:
        tau = ora.random_fr(nv, 1000 + nv)
        _srs[nv] = (zk.TrustedSetup.setup(tau), ora.kzg_multilinear_srs_g1(tau))
    srs, osrs = _srs[nv]
    v, z = ora.random_fr(1 << nv, rng.randrange(1 << 30)), ora.random_fr(nv, rng.randrange(1 << 30))
    mode = rng.random()          # a third each: folded levels derived per call / cached / cached with their shifted tables
    if mode < 0.33:
        srs.precompute_open()
    elif srs.level_tables is not None:
        srs.invalidate()
    proof = zk.MultilinearKZG.open(zk.Multilinear(v), z, srs, cache_folded_srs=mode < 0.66)
    want_ev, want_proofs = ora.kzg_open(v, z, osrs)
    ok = np.array_equal(proof.evaluation, want_ev)
    for got, w in zip(proof.proofs, want_proofs):
        a = ora.g1_to_affine(w)
        ok = ok and (bool(a[12]) == got.infinity) and (got.infinity or np.array_equal(got.xy, a[:12]))
    return ok, ("open", nv)


_usrs = {}
def case_uni_open():
    n = rng.randint(1, 200)
    if "s" not in _usrs:
        tau = ora.random_fr(1, 555)[0]
        _usrs["s"] = (zk.UnivariateKZG.generate_srs(tau, 200), ora.kzg_univariate_srs_g1(tau, 200))
    srs, osrs = _usrs["s"]
    c, z = ora.random_fr(n, rng.randrange(1 << 30)), ora.random_fr(1, rng.randrange(1 << 30))[0]
    proof = zk.UnivariateKZG.open(zk.DenseUnivariatePolynomial(c), z, srs)
    want_ev, want_proof = ora.univariate_kzg_open(c, z, osrs)
    a = ora.g1_to_affine(want_proof)
    ok = np.array_equal(proof.evaluation, want_ev) and (bool(a[12]) == proof.proof.infinity) and (proof.proof.infinity or np.array_equal(proof.proof.xy, a[:12]))
    com = zk.UnivariateKZG.commitment(zk.DenseUnivariatePolynomial(c), srs)
    w = ora.g1_to_affine(ora.kzg_commitment(c, osrs, False))
    return ok and np.array_equal(com.xy, w[:12]), ("uni_open", n)


def case_table_ops():
    la, lb = rng.randint(0, 7), rng.randint(0, 7)
    a, b = ora.random_fr(1 << la, rng.randrange(1 << 30)), ora.random_fr(1 << lb, rng.randrange(1 << 30))
    pa, pb = zk.Multilinear(a), zk.Multilinear(b)
    ok = np.array_equal(dev(pa.add_distinct(pb).evaluations), ora.mle_add_distinct(a, b))
    ok = ok and np.array_equal(dev(pa.mul_distinct(pb).evaluations), ora.mle_mul_distinct(a, b))
    k = rng.randint(0, 5)
    ok = ok and np.array_equal(dev(pa.add_to_front(k).evaluations), ora.mle_add_to_front(a, k))
    ok = ok and np.array_equal(dev(pa.add_to_back(k).evaluations), ora.mle_add_to_back(a, k))
    ok = ok and pa.to_bytes() == ora.mle_to_bytes(a)
    hs = ora.mle_half_sums(a) if la > 0 else None
    return ok, ("table_ops", la, lb, k)



def case_multifold():
    """partial_evaluations(points, [0] * k) -- the k-variable fold (VALU and MFMA forms, chunked weights, small-table form) -- against k
    single folds of the oracle"""
    log_n = rng.randint(4, 21 + BIG // 2)
    k = rng.randint(1, min(log_n - 1, 12))
    t = ora.random_fr(1 << log_n, rng.randrange(1 << 30))
    kind = rng.random()
    if kind < 0.15: t[:] = 0
    elif kind < 0.3: t = zk.Fr.from_ints([zk.Fr.MODULUS - 1] * 8)[rng.randrange(8)][None, :].repeat(1 << log_n, axis=0)   # r - 1 everywhere: largest limbs
    pts = ora.random_fr(k, rng.randrange(1 << 30))
    got = dev(zk.Multilinear(t).partial_evaluations(pts, [0] * k).evaluations)
    want = t
    for i in range(k):
        want = ora.mle_partial_evaluation(want, pts[i], 0)
    return np.array_equal(got, want), ("multifold", log_n, k)


def case_in_flight():
    """up to eight Sumcheck proofs in flight on lanes of their own, tables of mixed sizes"""
    depth = rng.randint(2, 8)
    logs = [rng.choice([3, 9, 12, 18, 19, 20, 21]) for _ in range(rng.randint(2, 10))]
    tabs = [ora.random_fr(1 << l, rng.randrange(1 << 30)) for l in logs]
    pend, got = [], []
    for t in tabs:
        sc = zk.Sumcheck(zk.Multilinear(t)); sc.poly_sum()
        pend.append(sc.prove_begin())
        if len(pend) == depth: got.append(pend.pop(0).wait())
    got += [h.wait() for h in pend]
    ok = True
    for t, (proof, ch) in zip(tabs, got):
        ws, wrp, wch = ora.sumcheck_prove(t)
        ok = ok and np.array_equal(proof.sum, ws) and np.array_equal(proof.univariate_poly, wrp) and np.array_equal(ch, wch)
    return ok, ("in_flight", depth, logs)


cases = [case_multi_k2, case_multifold, case_in_flight, case_open, case_uni_open, case_table_ops, case_sumcheck, case_fold_eval, case_composed, case_multi, case_commit, case_ntt, case_gkr, case_gkr_batch]
if len(sys.argv) > 3:
    cases = [c for c in cases if c.__name__ in sys.argv[3:]]
t0 = time.time()
while time.time() - t0 < budget:
    fn = rng.choice(cases)
    try:
        res = fn()
    except Exception as e:   # an exception is a failure too
        res = (False, (fn.__name__, "EXCEPTION %s: %s" % (type(e).__name__, e)))
    if res == "skip": continue
    ok, what = res
    counts[fn.__name__] = counts.get(fn.__name__, 0) + 1
    if not ok:
        fails += 1
        print("FAIL", what, flush=True)
print("cases:", counts, "failures:", fails)
sys.exit(min(fails, 100))

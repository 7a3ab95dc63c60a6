#!/usr/bin/env python3
"""Generates tests/golden/hot_path_vectors.json from tests/golden/model.py -- an independent pure-python model
(python ints + hashlib) written from the reference's Rust.  Neither `oracle/` nor the package is imported here.

The reference (Rust over arkworks) cannot be built or run in this image, and none of its tests pins a transcript
challenge, proof byte or commitment coordinate; so these vectors are MODEL-DERIVED, not reference-generated: inputs
are the reference's own unit-test inputs (cited per entry) plus hash-derived field elements, outputs come from the
model after `model.self_check()` pinned it on the value KATs the reference does hold.  tests/test_golden.py then
demands that the C oracle and the HIP path -- two further, separately written implementations -- reproduce every
entry bit for bit.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import model as M  # noqa: E402


def hx(vals):
    return ["%064x" % (v % M.R) for v in vals]


def pt(p):
    return {"x": "%096x" % (p[0] if p else 0), "y": "%096x" % (p[1] if p else 0), "inf": p is None}


def derived(tag, n):
    """n field elements from SHA-256 of a label: inputs anybody can regenerate without this repository"""
    return [int.from_bytes(hashlib.sha256(("zkhip-golden:%s:%d" % (tag, i)).encode()).digest(), "big") % M.R for i in range(n)]


def random_circuit(depth):
    """Circuit::random (circuit/src/circuit.rs:99-122)"""
    return [[("add" if li % 2 == 0 else "mul", (2 * g) % 2 ** (li + 1), (2 * g + 1) % 2 ** (li + 1)) for g in range(2 ** li)]
            for li in range(depth)]


GKR_1 = [[("mul", 0, 1)], [("add", 0, 1), ("mul", 2, 3)]]
GKR_2 = [[("add", 0, 1)],
         [("mul", 0, 1), ("add", 2, 3)],
         [("add", 0, 1), ("mul", 2, 3), ("mul", 4, 5), ("mul", 6, 7)],
         [("mul", 0, 1), ("mul", 2, 3), ("mul", 4, 5), ("add", 6, 7), ("mul", 8, 9), ("add", 10, 11), ("mul", 12, 13), ("mul", 14, 15)]]


def main():
    assert M.self_check()
    out = {"_about": "model-derived vectors (tests/golden/model.py: python ints + hashlib); see make_golden.py",
           "sumcheck": [], "composed": [], "multi_composed": [], "kzg": [], "kzg_open": [], "univariate_kzg_open": [],
           "ntt": [], "multiply": [], "gkr": []}

    for vals, src in [([0, 0, 2, 7, 3, 3, 6, 11], "sumcheck/src/sumcheck.rs:127-136"),
                      ([0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0], "sumcheck/src/sumcheck.rs:146-163"),
                      ([1, 3, 5, 7, 2, 4, 6, 8, 3, 5, 7, 9, 4, 6, 8, 10], "sumcheck/src/sumcheck.rs:175-192"),
                      (derived("sumcheck", 64), "hash-derived, 6 variables")]:
        s, rps, chs = M.sumcheck_prove([v % M.R for v in vals])
        out["sumcheck"].append({"source": src, "evals": hx(vals), "sum": hx([s])[0], "round_polys": [hx(r) for r in rps],
                                "challenges": hx(chs)})

    for tables, src in [([[3, 3, 5, 5], [0, 0, 0, 1]], "sumcheck/src/composed/composed_sumcheck.rs:152-161"),
                        ([[0, 0, 2, 7, 3, 3, 6, 11]], "sumcheck/src/composed/composed_sumcheck.rs:166-183"),
                        ([derived("composed-k3-%d" % k, 16) for k in range(3)], "hash-derived, K = 3, 4 variables"),
                        ([derived("composed-k5-%d" % k, 32) for k in range(5)],
                         "hash-derived, K = 5, 5 variables (the shape of sumcheck/benches/composed_sumcheck_benchmark.rs:33-78)")]:
        rps, chs = M.composed_prove([[v % M.R for v in t] for t in tables])
        out["composed"].append({"source": src, "tables": [hx(t) for t in tables], "round_polys": [hx(r) for r in rps],
                                "challenges": hx(chs)})

    p1, p2 = [0, 0, 0, 2], [0, 3, 0, 3]
    zero_mid = [[[1, 2, 3, 4], [0, 0, 0, 0]], [[5, 6, 7, 8]]]          # a term that interpolates to nothing, a kept zero sum
    cancel = [[[1, 1, 2, 2]], [[M.R - 1, M.R - 1, M.R - 2, M.R - 2]]]   # coefficients cancel in the merge-add: zeros are KEPT
    for terms, src in [([[p1], [p2]], "multi_composed_sumcheck.rs:217-232"),
                       ([[p1, p2], [p2, p1]], "multi_composed_sumcheck.rs:250-264"),
                       (zero_mid, "an all-zero table: its term's interpolation drops every coefficient (sparse_univariate.rs:55)"),
                       (cancel, "terms that cancel: the merge-add keeps zero sums (sparse_univariate.rs:159-203)"),
                       ([[derived("mc-a%d" % k, 16) for k in range(2)], [derived("mc-b%d" % k, 16) for k in range(3)]],
                        "hash-derived (2 + 3): sumcheck/benches/multi_composed_sumcheck_benchmark.rs:8-54's shape")]:
        terms = [[[v % M.R for v in t] for t in term] for term in terms]
        s = M.multi_composed_sum(terms)
        for partial in (False, True):
            rps, chs = M.multi_composed_prove(terms, s, partial)
            assert M.multi_composed_verify_partial(s, rps) is not None or not partial
            out["multi_composed"].append({"source": src, "terms": [[hx(t) for t in term] for term in terms], "partial": partial,
                                          "sum": hx([s])[0], "proof_bytes": M.proof_bytes(rps).hex(), "challenges": hx(chs)})

    kzg1 = ([0, 7, 0, 5, 0, 7, 4, 9], [2, 3, 4], [5, 9, 6])
    kzg2 = ([0, 0, 0, 2, 0, 0, 10, 12, 0, -12, 4, -6, 0, -12, 14, 4], [12, 9, 28, 40], [54, 90, 76, 160])
    for (vals, tau, _), src in [(kzg1, "kzg/src/multilinear_kzg.rs:133-148"), (kzg2, "kzg/src/multilinear_kzg.rs:151-177")]:
        c = M.commit([v % M.R for v in vals], M.multilinear_srs(tau), True)
        out["kzg"].append(dict(source=src, kind="multilinear", evals=hx(vals), tau=tau, **pt(c)))
    c = M.commit([1, 2, 3, 4, 5], M.univariate_srs(10, 4), False)
    out["kzg"].append(dict(source="kzg/src/univariate_kzg.rs:111-129", kind="univariate", coeffs=hx([1, 2, 3, 4, 5]), tau=10, max_degree=4, **pt(c)))
    for (vals, tau, z), src in [(kzg1, "kzg/src/multilinear_kzg.rs:131-155"), (kzg2, "kzg/src/multilinear_kzg.rs:157-197")]:
        ev, proofs = M.kzg_open([v % M.R for v in vals], z, M.multilinear_srs(tau))
        out["kzg_open"].append({"source": src, "evals": hx(vals), "tau": tau, "points": z, "evaluation": hx([ev])[0],
                                "proofs": [pt(p) for p in proofs]})
    ev, proof = M.univariate_open([1, 2, 3, 4, 5], 2, M.univariate_srs(10, 4))
    out["univariate_kzg_open"].append(dict(source="kzg/src/univariate_kzg.rs:111-129 (numerator poly - z, :70)", coeffs=hx([1, 2, 3, 4, 5]),
                                           tau=10, max_degree=4, z=2, evaluation=hx([ev])[0], proof=pt(proof)))

    for vec, size, src in [(list(range(1, 17)), 16, "Domain::new(16) (domain.rs:154-168 pins omega); input 1..16"),
                           (derived("ntt32", 32), 32, "hash-derived, Domain::new(32)"),
                           (derived("ntt-pad", 5), 8, "5 coefficients into Domain::new(5): zero padding (domain.rs:120-122)")]:
        out["ntt"].append({"source": src, "input": hx(vec), "size": size, "fft": hx(M.domain_fft(vec, size)), "ifft": hx(M.domain_ifft(vec, size))})
    a, b = derived("mul-a", 7), derived("mul-b", 4)
    out["multiply"].append({"source": "UnivariateEval::multiply (evaluation.rs:59-86), 7 x 4 coefficients", "a": hx(a), "b": hx(b),
                            "product": hx(M.univariate_multiply(a, b))})

    for layers, inp, src in [(GKR_1, [2, 3, 4, 5], "gkr/src/protocol.rs:209-232"),
                             (GKR_2, [2, 1, 3, 1, 4, 1, 2, 2, 3, 3, 4, 4, 2, 3, 3, 4], "gkr/src/protocol.rs:234-286"),
                             (random_circuit(3), derived("gkr-random3", 8), "Circuit::random(3) (circuit.rs:99-122), hash-derived inputs")]:
        ev = M.circuit_evaluation(layers, [v % M.R for v in inp])
        proof = M.gkr_prove(layers, ev)
        assert M.gkr_verify(layers, [v % M.R for v in inp], proof)
        out["gkr"].append({"source": src, "layers": [[list(g) for g in l] for l in layers], "input": hx(inp), "output": hx(ev[0]),
                           "w0": hx(proof["w0"]),
                           "proofs": [{"sum": hx([lp["sum"]])[0], "proof_bytes": M.proof_bytes(lp["rps"]).hex(),
                                       "challenges": hx(lp["challenges"]), "wb": hx([lp["wb"]])[0], "wc": hx([lp["wc"]])[0]}
                                      for lp in proof["layers"]]})
    with open(os.path.join(HERE, "hot_path_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()

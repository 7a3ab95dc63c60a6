#!/usr/bin/env python3
"""Generates tests/golden/hot_path_vectors.json.

The reference (Rust over arkworks) cannot be built or run in this image, and none of its tests pins a
transcript challenge, proof byte or commitment coordinate.  These vectors are therefore RESTATEMENT-DERIVED:
inputs are the reference's own unit-test inputs (cited per entry), outputs come from the CPU oracle after it
was pinned on every value-KAT the reference does hold (tests/test_oracle_kats.py) and cross-checked against
python ints + hashlib.  They freeze today's behaviour so that the oracle and the HIP path cannot drift apart
silently; they are not reference-generated goldens.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as ora  # noqa: E402

R = ora.R_MOD


def hx(arr):
    return ["%064x" % v for v in ora.fr_to_ints(arr)]


def main():
    out = {"_about": "restatement-derived vectors; see make_golden.py", "sumcheck": [], "multi_composed": [], "composed": [],
           "kzg": [], "kzg_open": [], "ntt": []}
    for vals, src in [([0, 0, 2, 7, 3, 3, 6, 11], "sumcheck/src/sumcheck.rs:127-136"),
                      ([0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0], "sumcheck/src/sumcheck.rs:146-163"),
                      ([1, 3, 5, 7, 2, 4, 6, 8, 3, 5, 7, 9, 4, 6, 8, 10], "sumcheck/src/sumcheck.rs:175-192")]:
        s, rp, ch = ora.sumcheck_prove(ora.fr_from_ints(vals))
        out["sumcheck"].append({"source": src, "evals": vals, "sum": hx(s)[0], "round_polys": [hx(r) for r in rp], "challenges": hx(ch)})
    for tables, src in [([[3, 3, 5, 5], [0, 0, 0, 1]], "sumcheck/src/composed/composed_sumcheck.rs:152-161"),
                        ([[0, 0, 2, 7, 3, 3, 6, 11]], "sumcheck/src/composed/composed_sumcheck.rs:166-183")]:
        t = np.stack([ora.fr_from_ints(v) for v in tables])
        rp, ch = ora.composed_prove(t)
        out["composed"].append({"source": src, "tables": tables, "round_polys": [hx(r) for r in rp], "challenges": hx(ch)})
    p1, p2 = [0, 0, 0, 2], [0, 3, 0, 3]
    for terms, src in [([[p1], [p2]], "multi_composed_sumcheck.rs:217-232"), ([[p1, p2], [p2, p1]], "multi_composed_sumcheck.rs:250-264")]:
        flat = np.stack([ora.fr_from_ints(t) for term in terms for t in term])
        sizes = [len(term) for term in terms]
        s = ora.multi_composed_sum(flat, sizes)
        for partial in (False, True):
            rps, ch = ora.multi_composed_prove(flat, sizes, s, partial)
            out["multi_composed"].append({"source": src, "terms": terms, "partial": partial, "sum": hx(s)[0],
                                          "proof_bytes": ora.multi_composed_proof_bytes(rps).hex(), "challenges": hx(ch)})
    for vals, tau, src in [([0, 7, 0, 5, 0, 7, 4, 9], [2, 3, 4], "kzg/src/multilinear_kzg.rs:133-148"),
                           ([0, 0, 0, 2, 0, 0, 10, 12, 0, -12, 4, -6, 0, -12, 14, 4], [12, 9, 28, 40], "kzg/src/multilinear_kzg.rs:151-177")]:
        c = ora.kzg_commitment(ora.fr_from_ints(vals), ora.kzg_multilinear_srs_g1(ora.fr_from_ints(tau)), True)
        x, y, inf = ora.g1_affine_ints(ora.g1_to_affine(c))
        out["kzg"].append({"source": src, "kind": "multilinear", "evals": vals, "tau": tau, "x": "%096x" % x, "y": "%096x" % y, "inf": inf})
    c = ora.kzg_commitment(ora.fr_from_ints([1, 2, 3, 4, 5]), ora.kzg_univariate_srs_g1(ora.fr_from_ints([10])[0], 4), False)
    x, y, inf = ora.g1_affine_ints(ora.g1_to_affine(c))
    out["kzg"].append({"source": "kzg/src/univariate_kzg.rs:111-129", "kind": "univariate", "coeffs": [1, 2, 3, 4, 5], "tau": 10,
                       "x": "%096x" % x, "y": "%096x" % y, "inf": inf})
    for vals, tau, z, src in [([0, 7, 0, 5, 0, 7, 4, 9], [2, 3, 4], [5, 9, 6], "kzg/src/multilinear_kzg.rs:131-155"),
                              ([0, 0, 0, 2, 0, 0, 10, 12, 0, -12, 4, -6, 0, -12, 14, 4], [12, 9, 28, 40], [54, 90, 76, 160],
                               "kzg/src/multilinear_kzg.rs:157-197")]:
        ev, proofs = ora.kzg_open(ora.fr_from_ints(vals), ora.fr_from_ints(z), ora.kzg_multilinear_srs_g1(ora.fr_from_ints(tau)))
        pts = []
        for pr in proofs:
            x, y, inf = ora.g1_affine_ints(ora.g1_to_affine(pr))
            pts.append({"x": "%096x" % x, "y": "%096x" % y, "inf": inf})
        out["kzg_open"].append({"source": src, "evals": vals, "tau": tau, "points": z, "evaluation": hx(ev.reshape(1, 4))[0], "proofs": pts})
    vec = list(range(1, 17))
    out["ntt"].append({"source": "Domain::new(16) (domain.rs:154-168 pins omega); input 1..16", "input": vec,
                       "fft": hx(ora.domain_fft(ora.fr_from_ints(vec), 16)), "ifft": hx(ora.domain_ifft(ora.fr_from_ints(vec), 16))})
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hot_path_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

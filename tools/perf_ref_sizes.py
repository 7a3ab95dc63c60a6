"""Diagnostic: the shapes of the reference's own criterion benches (all 2^8-sized) on the GPU path and on
the C oracle (one host core).  At these sizes every GPU entry point is launch-latency bound; the table in DESIGN.md comes from here."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk
from oracle import oracle as ora


def t_gpu(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def t_cpu(fn, reps=5):
    fn()
    t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e3


n = 256
tabs = [ora.random_fr(n, 10 + k) for k in range(5)]
dev = [zk.Multilinear(t) for t in tabs]
rows = []
# sumcheck_benchmark.rs: poly_sum + prove
def sc():
    s = zk.Sumcheck(dev[0]); s.poly_sum(); return s.prove()
rows.append(("Sumcheck poly_sum + prove, 2^8", t_gpu(sc), t_cpu(lambda: ora.sumcheck_prove(tabs[0]))))
for k in (2, 5):   # composed_sumcheck_benchmark.rs
    cm = zk.ComposedMultilinear(dev[:k])
    st = np.stack(tabs[:k])
    rows.append(("ComposedSumcheck::prove, %d tables x 2^8" % k, t_gpu(lambda: zk.ComposedSumcheck(cm).prove()), t_cpu(lambda: ora.composed_prove(st))))
terms = [zk.ComposedMultilinear(dev[:2]), zk.ComposedMultilinear(dev[2:5])]   # multi_composed_sumcheck_benchmark.rs: 2 + 3 tables
flat = np.stack(tabs)
s = zk.MultiComposedSumcheckProver.calculate_poly_sum(terms)
for name, fn, partial in (("prove", zk.MultiComposedSumcheckProver.prove, False), ("prove_partial", zk.MultiComposedSumcheckProver.prove_partial, True)):
    rows.append(("MultiComposedSumcheckProver::%s, (2 + 3) x 2^8" % name, t_gpu(lambda: fn(terms, s)),
                 t_cpu(lambda: ora.multi_composed_prove(flat, [2, 3], s, partial))))
# multilinear_kzg_benchmark.rs: commitment + open at 2^8
tau, z = ora.random_fr(8, 3), ora.random_fr(8, 4)
srs = zk.TrustedSetup.setup(tau)
osrs = ora.kzg_multilinear_srs_g1(tau)
rows.append(("MultilinearKZG::commitment, 2^8", t_gpu(lambda: zk.MultilinearKZG.commitment(dev[0], srs)), t_cpu(lambda: ora.kzg_commitment(tabs[0], osrs, True), 2)))
rows.append(("MultilinearKZG::open, 2^8 (8 proofs)", t_gpu(lambda: zk.MultilinearKZG.open(dev[0], z, srs)), t_cpu(lambda: ora.kzg_open(tabs[0], z, osrs), 1)))
# gkr_benchmark.rs: Circuit::random(8), 256 inputs
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gkr_cases import random_circuit
layers = random_circuit(8)
circuit = zk.Circuit.from_tuples(layers)
inp = ora.random_fr(256, 5)
ev = circuit.evaluation(inp)
oev = ora.circuit_evaluation(layers, inp)
rows.append(("GKRProtocol::prove, Circuit::random(8)", t_gpu(lambda: zk.GKRProtocol.prove(circuit, ev), 5), t_cpu(lambda: ora.gkr_prove(layers, oev), 1)))
print("| reference bench shape | GPU path (ms) | C oracle, one core (ms) |\n|---|---|---|")
for name, g, c in rows:
    print("| %s | %.3f | %.3f |" % (name, g, c))

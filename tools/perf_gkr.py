"""Diagnostic: GKRProtocol::prove on Circuit::random(depth) (the reference's gkr bench shape) on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk

for depth in [int(a) for a in sys.argv[1:]] or [4, 6, 8]:
    circuit = zk.Circuit.random(depth)
    inp = zk.Fr.random(2 ** depth, depth)
    ev = circuit.evaluation(inp)
    zk.GKRProtocol.prove(circuit, ev)
    torch.cuda.synchronize()
    reps = 3
    t = time.perf_counter()
    for _ in range(reps):
        proof = zk.GKRProtocol.prove(circuit, ev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("GKR prove depth %d (dense wiring table would be 2^%d, dense (b, c) tables 2^%d entries): %.2f ms, %d sumcheck proofs"
          % (depth, 3 * depth - 1, 2 * depth, dt * 1e3, len(proof.sumcheck_proofs)), flush=True)

"""Circuits and inputs of the reference's GKR / circuit tests, shared by the oracle (CPU) and HIP (GPU) suites.
Gates: (type, input_1, input_2); layer 0 is the output layer."""
A, M = "add", "mul"

# gkr/src/protocol.rs:209-232 (test_gkr_protocol_1) == circuit/src/circuit.rs:139-166 (test_circuit_evaluation_1)
GKR_1 = dict(layers=[[(M, 0, 1)], [(A, 0, 1), (M, 2, 3)]], input=[2, 3, 4, 5], evaluation=[[100], [5, 20], [2, 3, 4, 5]])

# gkr/src/protocol.rs:234-286 (test_gkr_protocol_2): output 224
GKR_2 = dict(layers=[[(A, 0, 1)],
                     [(M, 0, 1), (A, 2, 3)],
                     [(A, 0, 1), (M, 2, 3), (M, 4, 5), (M, 6, 7)],
                     [(M, 0, 1), (M, 2, 3), (M, 4, 5), (A, 6, 7), (M, 8, 9), (A, 10, 11), (M, 12, 13), (M, 14, 15)]],
             input=[2, 1, 3, 1, 4, 1, 2, 2, 3, 3, 4, 4, 2, 3, 3, 4], output=224)

# circuit/src/circuit.rs:209-260 (test_circuit_evaluation_3), wiring tests :262-518 use the same circuit
CIRCUIT_3 = dict(layers=[[(A, 0, 1)], [(A, 0, 1), (M, 2, 3)], [(A, 0, 1), (M, 2, 3), (M, 4, 5), (M, 6, 7)]],
                 input=[2, 3, 1, 4, 1, 2, 3, 4], evaluation=[[33], [9, 24], [5, 4, 2, 12], [2, 3, 1, 4, 1, 2, 3, 4]])

# circuit/src/circuit.rs:168-207 (test_circuit_evaluation_2): two output gates
CIRCUIT_2 = dict(layers=[[(M, 0, 1), (M, 2, 3)], [(M, 0, 0), (M, 1, 1), (M, 1, 2), (M, 3, 3)]],
                 input=[3, 2, 3, 1], evaluation=[[36, 6], [9, 4, 6, 1], [3, 2, 3, 1]])


def random_circuit(num_of_layers):
    """Circuit::random (circuit.rs:99-122): layer i has 2^i gates over 2^(i+1) inputs, Add on even layers, Mul on odd"""
    layers = []
    for li in range(num_of_layers):
        n_in = 2 ** (li + 1)
        layers.append([(A if li % 2 == 0 else M, (2 * g) % n_in, (2 * g + 1) % n_in) for g in range(2 ** li)])
    return layers


def scrambled_circuit(num_of_layers, seed):
    """Same shape as Circuit::random, but every gate draws its type and both inputs at random: inputs shared between gates
    (several gates on one (b, c) pair, values read twice), b == c, unused inputs -- what Circuit::random never produces."""
    import random
    rng = random.Random(seed)
    layers = []
    for li in range(num_of_layers):
        n_in = 2 ** (li + 1)
        layers.append([(rng.choice((A, M)), rng.randrange(n_in), rng.randrange(n_in)) for _ in range(2 ** li)])
    return layers


def gkr_proof_mismatches(ora, proof, want):
    """Field by field: a GKRProof of the HIP prover (zk.GKRProtocol.prove / prove_sharded) against the oracle's
    (ora.gkr_prove / ora.gkr_prove_sparse) -- w_0, and per layer the claimed sum, ComposedSumcheckProof::to_bytes, the challenges
    prove_partial returned, w_b and w_c.  Returns the list of what differs (empty = bit-identical)."""
    import numpy as np
    bad = []
    if len(proof.sumcheck_proofs) != want.n_proofs:
        return ["n_proofs %d != %d" % (len(proof.sumcheck_proofs), want.n_proofs)]
    w0 = proof.w_0_mle.evaluations
    w0 = (w0.cpu().numpy() if hasattr(w0, "cpu") else np.asarray(w0)).view(np.uint64).reshape(-1)
    if [int(v) for v in w0] != list(want.w0[0:8]):
        bad.append("w_0")
    for k, sp in enumerate(proof.sumcheck_proofs):
        w_sum, w_rps, w_wb, w_wc = want.layer(k)
        if not np.array_equal(sp.sum, w_sum):
            bad.append("layer %d sum" % k)
        if sp.to_bytes() != ora.multi_composed_proof_bytes(w_rps):
            bad.append("layer %d proof bytes" % k)
        if not np.array_equal(np.asarray(proof._challenges[k]).reshape(-1, 4), want.layer_challenges(k)):
            bad.append("layer %d challenges" % k)
        if not (np.array_equal(proof.wb_s[k], w_wb) and np.array_equal(proof.wc_s[k], w_wc)):
            bad.append("layer %d w_b / w_c" % k)
    return bad

"""circuit::{Gate, CircuitLayer, Circuit} and gkr::GKRProtocol::prove with every table in HBM.

Mirrors circuit/src/circuit.rs:8-122, circuit/src/gate.rs, gkr/src/utils.rs:8-56 and gkr/src/protocol.rs:10-117: the
caller of the hot path (wiring tables -> folds at r_b / r_c -> outer sum / product of the layer values -> the
multi-composed sumcheck prover -> evaluations of w at b and c).  GKRProtocol.prove is ONE C-ABI call: every table operation, every
sumcheck and the outer Fiat-Shamir transcript (a few hundred bytes per layer, absorbed by a hasher workgroup beside the closing kernels,
csrc/gkr.hip) run on the device; only prove_stepwise -- the mirror that makes one call per reference line -- keeps its outer transcript
on the host (hashlib).  The verifier is host-side in the reference and out of scope; tests check proofs with the oracle's
restated verifier.
"""
import ctypes as C
import hashlib
import threading

import numpy as np

from zk_cryptography_amd import _native as N
from zk_cryptography_amd.composed import ComposedMultilinear, MultiComposedSumcheckProver
from zk_cryptography_amd.field import Fr, R_MOD
from zk_cryptography_amd.polynomial import Multilinear

ADD, MUL = "add", "mul"


class Gate:
    """circuit/src/gate.rs:7-17.  Immutable: a gate is changed by REPLACING it in its layer (`layer.layer[i] = Gate(...)`), which
    the layer's gate list records -- that is how a circuit edited in place reaches the device copy (GKRProtocol._device_circuit)."""

    __slots__ = ("gate_type", "inputs")

    def __init__(self, gate_type, inputs):
        assert gate_type in (ADD, MUL)
        object.__setattr__(self, "gate_type", gate_type)
        object.__setattr__(self, "inputs", (int(inputs[0]), int(inputs[1])))

    def __setattr__(self, name, value):
        raise AttributeError("Gate is immutable: replace it in its layer (layer.layer[i] = Gate(...))")

    __delattr__ = __setattr__


class _GateList(list):
    """A layer's gates: a list that counts its mutations (`version`), so that the arrays and the device copy derived from it are
    never stale (Python lists carry no version counter of their own; hashing 2^20 gates per proof would cost more than the proof)."""

    version = 0

    def _touch(self):
        self.version += 1


def _counting(name):
    base = getattr(list, name)

    def method(self, *a, **k):
        self._touch()
        return base(self, *a, **k)
    method.__name__ = name
    return method


for _m in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove", "clear", "reverse", "sort"):
    setattr(_GateList, _m, _counting(_m))


class CircuitLayer:
    """circuit/src/circuit.rs:8-22"""

    def __init__(self, layer):
        self.layer = layer

    @property
    def layer(self):
        return self._layer

    @layer.setter
    def layer(self, gates):
        self._layer = _GateList(gates)
        self._arr = None

    def _arrays(self):
        """(gate_type u8, in0 u32, in1 u32) as the C ABI takes them; built once per state of the gate list (a layer of 2^19 gates
        is 1.5 M Python objects)"""
        if getattr(self, "_arr", None) is None or self._arr_version != self._layer.version:
            gt = np.array([0 if g.gate_type == ADD else 1 for g in self._layer], dtype=np.uint8)
            i0 = np.array([g.inputs[0] for g in self._layer], dtype=np.uint32)
            i1 = np.array([g.inputs[1] for g in self._layer], dtype=np.uint32)
            for a in (gt, i0, i1):
                a.setflags(write=False)
            self._arr = (gt, i0, i1)
            self._arr_version = self._layer.version
        return self._arr


class Circuit:
    """circuit/src/circuit.rs:13-122; layers[0] is the output layer"""

    def __init__(self, layers):
        self.layers = list(layers)

    def _stamp(self):
        """identity and mutation count of every layer's gate list: what a derived device copy is valid for"""
        return tuple((layer, layer.layer, layer.layer.version) for layer in self.layers)

    @staticmethod
    def from_tuples(layers):
        """[[(type, in0, in1), ...], ...] -> Circuit"""
        return Circuit([CircuitLayer([Gate(t, (a, b)) for t, a, b in layer]) for layer in layers])

    @staticmethod
    def random(num_of_layers):
        """Circuit::random (circuit.rs:99-122)"""
        layers = []
        for li in range(num_of_layers):
            n_in = 2 ** (li + 1)
            layers.append(CircuitLayer([Gate(ADD if li % 2 == 0 else MUL, ((2 * g) % n_in, (2 * g + 1) % n_in))
                                        for g in range(2 ** li)]))
        return Circuit(layers)

    def evaluation(self, inp):
        """Circuit::evaluation (circuit.rs:31-57) -> list of device tables (int64 [len, 4]), output layer first, input last.
        Layers need not be powers of two here (the reference only requires that of tables it turns into Multilinears)."""
        import torch
        from zk_cryptography_amd.polynomial import _to_device
        cur = _to_device(inp, None)
        ctx = N.Context.get(cur.device.index)
        layers = [cur]
        for layer in reversed(self.layers):
            gt, i0, i1 = layer._arrays()
            out = torch.empty((len(layer.layer), 4), dtype=torch.int64, device=cur.device)
            N.check(N.lib().zkhip_circuit_layer_eval(ctx.handle, N.ptr(cur), C.c_size_t(cur.shape[0]), gt.ctypes.data_as(C.c_void_p),
                                                     i0.ctypes.data_as(C.c_void_p), i1.ctypes.data_as(C.c_void_p),
                                                     C.c_size_t(len(layer.layer)), N.ptr(out)), "circuit_layer_eval")
            layers.append(out)
            cur = out
        layers.reverse()
        return layers

    def add_mult_mle(self, layer_index):
        """Circuit::add_mult_mle (circuit.rs:59-97) -> (add_mle, mul_mle) as device Multilinears"""
        import torch
        N.lib().zkhip_gkr_mle_size.restype = C.c_size_t
        size = N.lib().zkhip_gkr_mle_size(C.c_uint32(layer_index))
        gt, i0, i1 = self.layers[layer_index]._arrays()
        add = torch.empty((size, 4), dtype=torch.int64, device="cuda")
        mul = torch.empty((size, 4), dtype=torch.int64, device="cuda")
        ctx = N.Context.get(add.device.index)
        N.check(N.lib().zkhip_circuit_add_mult_mle(ctx.handle, gt.ctypes.data_as(C.c_void_p), i0.ctypes.data_as(C.c_void_p),
                                                   i1.ctypes.data_as(C.c_void_p), C.c_size_t(len(gt)), C.c_uint32(layer_index),
                                                   N.ptr(add), N.ptr(mul)), "circuit_add_mult_mle")
        return Multilinear(add), Multilinear(mul)


class FiatShamirTranscript:
    """transcripts/fiat-shamir/src/fiat_shamir.rs:10-40 on the host (the GKR outer transcript: a few hundred bytes per layer)"""

    def __init__(self):
        self.hasher = hashlib.sha256()

    def commit(self, new_data):
        self.hasher.update(new_data)

    def challenge(self):
        response = self.hasher.digest()
        self.hasher = hashlib.sha256()
        self.hasher.update(response)
        return response

    def evaluate_challenge_into_field(self):
        """F::from_be_bytes_mod_order -> Montgomery limbs uint64[4]"""
        return Fr.from_int(int.from_bytes(self.challenge(), "big") % R_MOD)

    def evaluate_n_challenge_into_field(self, n):
        return np.stack([self.evaluate_challenge_into_field() for _ in range(n)]) if n else np.zeros((0, 4), dtype=np.uint64)


class DeviceFiatShamirTranscript:
    """The same transcript with its hash on the GPU (zkhip_transcript_challenge: the device code every prover's transcript runs --
    schedule on the sixteen lanes of a row, state rounds on six), for holding that hash against an independent SHA-256.  commit()
    gathers bytes on the host; challenge() hashes `digest of the previous challenge || committed bytes` on the device."""

    def __init__(self, device=None):
        self._prefix = None
        self._pending = bytearray()
        self._ctx = N.Context.get(device)

    def commit(self, new_data):
        self._pending += bytes(new_data)

    def challenge(self):
        data = bytes(self._pending)
        out = (C.c_uint8 * 32)()
        pre = (C.c_uint8 * 32).from_buffer_copy(self._prefix) if self._prefix is not None else None
        buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
        N.check(N.lib().zkhip_transcript_challenge(self._ctx.handle, pre, buf, C.c_size_t(len(data)), out), "transcript_challenge")
        self._prefix = bytes(out)
        self._pending = bytearray()
        return self._prefix

    def evaluate_challenge_into_field(self):
        return Fr.from_int(int.from_bytes(self.challenge(), "big") % R_MOD)

    def evaluate_n_challenge_into_field(self, n):
        return np.stack([self.evaluate_challenge_into_field() for _ in range(n)]) if n else np.zeros((0, 4), dtype=np.uint64)


class GKRProof:
    """gkr/src/protocol.rs:10-15"""

    def __init__(self, sumcheck_proofs, wb_s, wc_s, w_0_mle):
        self.sumcheck_proofs = sumcheck_proofs
        self.wb_s = wb_s
        self.wc_s = wc_s
        self.w_0_mle = w_0_mle


def _fmul(a, b):
    return Fr.from_int(Fr.to_ints(a)[0] * Fr.to_ints(b)[0] % R_MOD)


def _fadd(a, b):
    return Fr.from_int((Fr.to_ints(a)[0] + Fr.to_ints(b)[0]) % R_MOD)


_DEVICE_CIRCUITS_LOCK = threading.Lock()      # guards the per-Circuit lists of device copies (GKRProtocol._device_circuit)


class _DeviceCircuit:
    """zkhip_circuit handle (include/zkhip.h): the circuit's gate arrays and CSR groupings resident in HBM.  The CONTEXT owns it
    (Context._circuits; a zkhip_circuit refers to its zkhip_ctx, so Context.destroy() destroys it first); the Circuit only lists it and
    refers to the context weakly -- a worker thread's context, with its workspace, goes when the thread does, not when the Circuit does."""

    def __init__(self, ctx, circuit, shape):
        import weakref
        arrays = [layer._arrays() for layer in circuit.layers]
        gt = np.concatenate([a[0] for a in arrays])
        i0 = np.concatenate([a[1] for a in arrays])
        i1 = np.concatenate([a[2] for a in arrays])
        self._ctx, self.shape = weakref.ref(ctx), shape
        self.handle = C.c_void_p()
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        st = N.lib().zkhip_circuit_create(ctx.handle, C.c_uint32(len(shape)), (C.c_size_t * len(shape))(*shape), p(gt), p(i0), p(i1),
                                          C.byref(self.handle))
        N.check(st, "circuit_create")
        ctx._circuits.append(self)

    @property
    def ctx(self):
        return self._ctx()

    def alive(self):
        ctx = self._ctx()
        return bool(self.handle) and ctx is not None and bool(ctx.handle)

    def close(self):
        """zkhip_circuit_destroy; called by Context.destroy() BEFORE the context goes (the handle dereferences it)"""
        h, self.handle = self.handle, None
        if h:
            N.lib().zkhip_circuit_destroy(h)
        ctx = self._ctx()
        if ctx is not None and self in ctx._circuits:
            ctx._circuits.remove(self)

    def __del__(self):
        try:
            if self.alive():
                self.close()
        except Exception:
            pass


class GKRProtocol:
    @staticmethod
    def _layer_sumcheck(add_bc, mul_bc, w_mle, claimed_sum, transcript, proof):
        """The shared tail of generate_layer_one_prove_sumcheck (gkr/src/utils.rs:27-55) and the loop body of
        GKRProtocol::prove (protocol.rs:78-107): returns (claimed_sum, alpha, beta, r_b, r_c)."""
        wb, wc = w_mle, w_mle
        wb_add_wc = wb.add_distinct(wc)
        wb_mul_wc = wb.mul_distinct(wc)
        fbc_add = ComposedMultilinear([add_bc, wb_add_wc])
        fbc_mul = ComposedMultilinear([mul_bc, wb_mul_wc])
        sumcheck_proof, challenges = MultiComposedSumcheckProver.prove_partial([fbc_add, fbc_mul], claimed_sum)
        transcript.commit(sumcheck_proof.to_bytes())
        proof.sumcheck_proofs.append(sumcheck_proof)
        half = len(challenges) // 2
        b, c = challenges[:half], challenges[half:]
        eval_wb, eval_wc = wb.evaluation(b), wc.evaluation(c)
        proof.wb_s.append(eval_wb)
        proof.wc_s.append(eval_wc)
        alpha = transcript.evaluate_challenge_into_field()
        beta = transcript.evaluate_challenge_into_field()
        return _fadd(_fmul(alpha, eval_wb), _fmul(beta, eval_wc)), alpha, beta, b, c

    @staticmethod
    def prove(circuit, circuit_evaluation):
        """GKRProtocol::prove (protocol.rs:21-117) through the single C-ABI entry point zkhip_gkr_prove;
        circuit_evaluation as returned by Circuit.evaluation (device tables)."""
        from zk_cryptography_amd.composed import MAX_MONO, MultiComposedSumcheckProof
        nl = len(circuit.layers)
        assert len(circuit_evaluation) == nl + 1
        tables = [t.contiguous() for t in circuit_evaluation]
        ptrs = (C.c_void_p * (nl + 1))(*[t.data_ptr() for t in tables])
        lens = (C.c_size_t * (nl + 1))(*[t.shape[0] for t in tables])
        stride = 2 * nl
        sums = np.zeros((nl, 4), dtype=np.uint64)
        n_rounds = np.zeros(nl, dtype=np.uint32)
        rp_lens = np.zeros((nl, stride), dtype=np.uint32)
        rps = np.zeros((nl, stride, MAX_MONO, 2, 4), dtype=np.uint64)
        wb, wc = np.zeros((nl, 4), dtype=np.uint64), np.zeros((nl, 4), dtype=np.uint64)
        w0 = np.zeros((2, 4), dtype=np.uint64)
        ctx = N.Context.get(tables[0].device.index)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        chal = np.zeros((nl, stride, 4), dtype=np.uint64)
        # the circuit lives on the device: gate arrays and their groupings are validated, built and uploaded once per Circuit content
        dev = GKRProtocol._device_circuit(circuit, ctx)
        st = N.lib().zkhip_gkr_prove_circuit(dev.handle, ptrs, lens, p(sums), p(n_rounds), p(rp_lens), p(rps), p(wb), p(wc), p(w0), p(chal))
        N.check(st, "gkr_prove: every layer must hold a power-of-two number of values, 2^l gates in layer l")
        # the arrays above belong to this proof alone: the per-layer proofs are views into them (round polynomials are
        # unpacked into SparseUnivariatePolynomial objects on first use)
        proofs = [MultiComposedSumcheckProof.from_packed(rps[k, : n_rounds[k]], rp_lens[k, : n_rounds[k]], sums[k]) for k in range(nl)]
        proof = GKRProof(proofs, list(wb), list(wc), Multilinear(w0))
        proof._challenges = [chal[k, : n_rounds[k]] for k in range(nl)]   # not part of the reference's struct
        return proof

    @staticmethod
    def prove_batch(circuit, circuit_evaluations, max_lanes=0):
        """One GKRProtocol::prove per entry of circuit_evaluations (each as Circuit.evaluation returns it) in ONE C-ABI call,
        zkhip_gkr_prove_batch: the proofs run side by side on the context's internal lanes (a proof is a chain of small dependent
        kernels; gkr/benches/gkr_benchmark.rs:11-27 proves input after input).  Returns the proofs GKRProtocol.prove would, in order."""
        from zk_cryptography_amd.composed import MAX_MONO, MultiComposedSumcheckProof
        nl, B = len(circuit.layers), len(circuit_evaluations)
        if B == 0:
            return []
        tables = [[t.contiguous() for t in ev] for ev in circuit_evaluations]
        assert all(len(ev) == nl + 1 for ev in tables)
        lens0 = [t.shape[0] for t in tables[0]]
        assert all([t.shape[0] for t in ev] == lens0 for ev in tables), "one circuit: every evaluation has the same layer sizes"
        ptrs = (C.c_void_p * (B * (nl + 1)))(*[t.data_ptr() for ev in tables for t in ev])
        lens = (C.c_size_t * (nl + 1))(*lens0)
        stride = 2 * nl
        sums = np.zeros((B, nl, 4), dtype=np.uint64)
        n_rounds = np.zeros((B, nl), dtype=np.uint32)
        rp_lens = np.zeros((B, nl, stride), dtype=np.uint32)
        rps = np.zeros((B, nl, stride, MAX_MONO, 2, 4), dtype=np.uint64)
        wb, wc = np.zeros((B, nl, 4), dtype=np.uint64), np.zeros((B, nl, 4), dtype=np.uint64)
        w0 = np.zeros((B, 2, 4), dtype=np.uint64)
        chal = np.zeros((B, nl, stride, 4), dtype=np.uint64)
        status = np.zeros(B, dtype=np.int32)
        ctx = N.Context.get(tables[0][0].device.index)
        dev = GKRProtocol._device_circuit(circuit, ctx)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        st = N.lib().zkhip_gkr_prove_batch(dev.handle, C.c_uint32(B), C.c_uint32(max_lanes), ptrs, lens, p(sums), p(n_rounds), p(rp_lens), p(rps), p(wb), p(wc),
                                           p(w0), p(chal), p(status))
        if st == N.ERR_HIP:
            raise N.ZkhipError("gkr_prove_batch: HIP runtime error (hipError %d); per-proof statuses %s" % (N.lib().zkhip_last_hip_error(ctx.handle), status.tolist()), st)
        N.check(st, "gkr_prove_batch: every layer must hold a power-of-two number of values, 2^l gates in layer l")
        out = []
        for b in range(B):
            proofs = [MultiComposedSumcheckProof.from_packed(rps[b, k, : n_rounds[b, k]], rp_lens[b, k, : n_rounds[b, k]], sums[b, k]) for k in range(nl)]
            proof = GKRProof(proofs, list(wb[b]), list(wc[b]), Multilinear(w0[b]))
            proof._challenges = [chal[b, k, : n_rounds[b, k]] for k in range(nl)]
            out.append(proof)
        return out

    @staticmethod
    def _device_circuit(circuit, ctx):
        """The circuit resident in HBM (zkhip_circuit): gate arrays and their groupings validated, built and uploaded once per
        Circuit STATE -- gates are immutable and every layer's gate list counts its mutations, so a gate replaced in place proves the edited circuit
        (the reference reads &Circuit on every call, gkr/src/protocol.rs:21-25).  The key holds the layer objects themselves
        (identity, not id(): no address reuse) and their gate lists' mutation counters."""
        shape = [len(layer.layer) for layer in circuit.layers]
        stamp = circuit._stamp()

        def fresh(dev):
            return dev.shape == shape and len(dev.stamp) == len(stamp) and \
                all(a[0] is b[0] and a[1] is b[1] and a[2] == b[2] for a, b in zip(dev.stamp, stamp))
        # one device copy per CONTEXT (contexts are per host thread: several threads may prove one Circuit at once, each on its own
        # stream); copies of an older circuit state are dropped
        with _DEVICE_CIRCUITS_LOCK:
            old = list(getattr(circuit, "_devices", ()))
            devs = [d for d in old if d.alive() and fresh(d)]
            for d in old:
                if d.alive() and not fresh(d):
                    d.close()                      # a copy of an older circuit state: give its device memory back now
            dev = next((d for d in devs if d.ctx is ctx), None)
            if dev is None:
                dev = _DeviceCircuit(ctx, circuit, shape)
                dev.stamp = stamp
                devs.append(dev)
            circuit._devices = devs
            circuit._device = dev                  # (the copy of the last call: what the tests look at)
        return dev

    @staticmethod
    def prove_sharded(circuit, circuit_evaluation, world=1, rank=0, group=None, dist=None, use_stages=None, comm=None):
        """GKRProtocol::prove (protocol.rs:21-117) with every layer's tables and sumcheck SHARDED over `world` ranks (SURVEY 8e,
        "GKR tables"; BASELINE configs[3]: "evals sharded across 8") through the single C-ABI entry point zkhip_gkr_prove_sharded:
        rank g builds ONLY rows j * world + g of the layer's seven linear-size sumcheck tables and of the layer's values, and the
        rounds over b and over c run on those shards, TWO rounds per exchange (use_stages: default on for world > 1), every
        exchange stream-ordered inside the library.  What the rows gather from by wire index -- the layer's values, the gate
        weights, eq(u) -- stays whole on every rank (random wiring reads any of them).  Layers narrower than 2 * world values run
        unsharded on every rank.  Returns the proof GKRProtocol.prove returns, bit for bit, on every rank; the number of
        collectives is left in proof._exchanges."""
        from zk_cryptography_amd import distributed as D
        from zk_cryptography_amd.composed import MAX_MONO, MultiComposedSumcheckProof
        nl = len(circuit.layers)
        assert len(circuit_evaluation) == nl + 1
        tables = [t.contiguous() for t in circuit_evaluation]
        ctx = N.Context.get(tables[0].device.index)
        dev = GKRProtocol._device_circuit(circuit, ctx)
        if comm is None:
            comm = D.Comm.get(ctx, world, rank, dist, group)
        ptrs = (C.c_void_p * (nl + 1))(*[t.data_ptr() for t in tables])
        lens = (C.c_size_t * (nl + 1))(*[t.shape[0] for t in tables])
        stride = 2 * nl
        sums = np.zeros((nl, 4), dtype=np.uint64)
        n_rounds = np.zeros(nl, dtype=np.uint32)
        rp_lens = np.zeros((nl, stride), dtype=np.uint32)
        rps = np.zeros((nl, stride, MAX_MONO, 2, 4), dtype=np.uint64)
        wb, wc = np.zeros((nl, 4), dtype=np.uint64), np.zeros((nl, 4), dtype=np.uint64)
        w0 = np.zeros((2, 4), dtype=np.uint64)
        chal = np.zeros((nl, stride, 4), dtype=np.uint64)
        ex = C.c_uint32(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        st = N.lib().zkhip_gkr_prove_sharded(dev.handle, comm.handle, ptrs, lens, C.c_int(-1 if use_stages is None else int(bool(use_stages))),
                                             p(sums), p(n_rounds), p(rp_lens), p(rps), p(wb), p(wc), p(w0), p(chal), C.byref(ex))
        comm.check(st, "gkr_prove_sharded: every layer must hold a power-of-two number of values, 2^l gates in layer l")
        proofs = [MultiComposedSumcheckProof.from_packed(rps[k, : n_rounds[k]], rp_lens[k, : n_rounds[k]], sums[k]) for k in range(nl)]
        proof = GKRProof(proofs, list(wb), list(wc), Multilinear(w0))
        proof._challenges = [chal[k, : n_rounds[k]] for k in range(nl)]
        proof._exchanges = ex.value
        return proof

    @staticmethod
    def prove_stepwise(circuit, circuit_evaluation):
        """The same prover spelled out call by call over the mirror's types, line for line with protocol.rs:21-117
        (kept as a cross-check of zkhip_gkr_prove and as the reading order of the reference)."""
        import torch
        transcript = FiatShamirTranscript()
        ev0 = circuit_evaluation[0]
        pad = torch.zeros((1, 4), dtype=torch.int64, device=ev0.device)
        w_0_mle = Multilinear(torch.cat([ev0, pad]))              # Multilinear::new panics unless 2^k entries
        proof = GKRProof([], [], [], w_0_mle)
        transcript.commit(w_0_mle.to_bytes())
        n_r = transcript.evaluate_n_challenge_into_field(w_0_mle.n_vars)
        claimed_sum = w_0_mle.evaluation(n_r)

        add_mle_1, mult_mle_1 = circuit.add_mult_mle(0)
        w_1_mle = Multilinear(circuit_evaluation[1])
        zeros = [0] * len(n_r)
        add_rbc = add_mle_1.partial_evaluations(n_r, zeros)
        mul_rbc = mult_mle_1.partial_evaluations(n_r, zeros)
        claimed_sum, alpha, beta, r_b, r_c = GKRProtocol._layer_sumcheck(add_rbc, mul_rbc, w_1_mle, claimed_sum, transcript, proof)

        for layer_index in range(2, len(circuit_evaluation)):
            add_mle, mult_mle = circuit.add_mult_mle(layer_index - 1)
            zeros = [0] * len(r_b)
            add_rb_bc = add_mle.partial_evaluations(r_b, zeros)
            mul_rb_bc = mult_mle.partial_evaluations(r_b, zeros)
            add_rc_bc = add_mle.partial_evaluations(r_c, zeros)
            mul_rc_bc = mult_mle.partial_evaluations(r_c, zeros)
            w_i_mle = Multilinear(circuit_evaluation[layer_index])
            add_alpha_beta = (add_rb_bc * alpha) + (add_rc_bc * beta)
            mul_alpha_beta = (mul_rb_bc * alpha) + (mul_rc_bc * beta)
            claimed_sum, alpha, beta, r_b, r_c = GKRProtocol._layer_sumcheck(add_alpha_beta, mul_alpha_beta, w_i_mle, claimed_sum,
                                                                            transcript, proof)
        return proof


class SuccintGKRProof(GKRProof):
    """gkr/src/succint_protocol.rs:21-29: GKRProof + the two KZG openings of the (blown-up) input layer"""

    def __init__(self, base, proof_wb_opening, proof_wc_opening):
        super().__init__(base.sumcheck_proofs, base.wb_s, base.wc_s, base.w_0_mle)
        self._challenges = base._challenges
        self.proof_wb_opening = proof_wb_opening
        self.proof_wc_opening = proof_wc_opening


class SuccintGKRProtocol:
    @staticmethod
    def prove(circuit, circuit_evaluation, tau):
        """SuccintGKRProtocol::prove (succint_protocol.rs:36-167) -> (commitment, proof).  The sumcheck part is
        GKRProtocol::prove verbatim (:38-131 repeat protocol.rs:21-108); at the input layer the values are blown up
        to the SRS size (add_to_back, :134-137), committed (:148) and opened at b || 0.. and c || 0.. (:139-151)."""
        from zk_cryptography_amd.kzg import MultilinearKZG
        assert len(circuit_evaluation) >= 3, "the commitment is made inside `for layer_index in 2..len` (:82-152)"
        base = GKRProtocol.prove(circuit, circuit_evaluation)
        w_i_mle = Multilinear(circuit_evaluation[-1])
        srs_vars = (len(tau)).bit_length() - 1
        assert 1 << srs_vars == len(tau), "Value is not a power of 2"        # gkr/src/utils.rs:100-111 exponent()
        blow_up = srs_vars - w_i_mle.n_vars                                   # usize underflow panics in the reference
        assert blow_up >= 0
        poly = w_i_mle.add_to_back(blow_up)
        ch = base._challenges[-1]
        half = len(ch) // 2
        zeros = np.zeros((poly.n_vars - half, 4), dtype=np.uint64)
        b_clone = np.concatenate([ch[:half], zeros])
        c_clone = np.concatenate([ch[half:], np.zeros((poly.n_vars - (len(ch) - half), 4), dtype=np.uint64)])
        commitment = MultilinearKZG.commitment(poly, tau)
        return commitment, SuccintGKRProof(base, MultilinearKZG.open(poly, b_clone, tau), MultilinearKZG.open(poly, c_clone, tau))

"""ms per proof of zkhip_gkr_prove_batch by (depth, proofs per call, lanes), straight through the C ABI (outputs preallocated once: no
Python work per proof inside the timed region).  usage (GPU box): python tools/perf_gkr_batch.py [depth:B:lanes ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk
from zk_cryptography_amd import _native as N
from zk_cryptography_amd.gkr import GKRProtocol

if os.environ.get("GKR_BATCH_AFTER_IN_FLIGHT"):        # Sumcheck proofs in flight first: their lanes' streams exist when the batch's lanes come into being
    _polys = [zk.Multilinear(torch.randint(0, 2 ** 62, (1 << 24, 4), dtype=torch.int64, device="cuda")) for _ in range(8)]
    _pend = []
    for _i in range(int(os.environ["GKR_BATCH_AFTER_IN_FLIGHT"]) * 3):
        _sc = zk.Sumcheck(_polys[_i % 8]); _sc.poly_sum(); _pend.append(_sc.prove_begin())
        if len(_pend) == int(os.environ["GKR_BATCH_AFTER_IN_FLIGHT"]): _pend.pop(0).wait()
    for _h in _pend: _h.wait()
    torch.cuda.synchronize()
    del _polys
cases = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(8, 8, 8), (8, 32, 8), (8, 32, 4), (8, 32, 2), (8, 32, 1), (20, 8, 8), (20, 8, 4), (20, 8, 2), (20, 8, 1)]
p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
circuits = {}
for depth, B, lanes in cases:
    if depth not in circuits:
        circuit = zk.Circuit.random(depth)
        evs = [circuit.evaluation(zk.Fr.synthetic(2 ** depth, 4000 + b)) for b in range(max(b_ for d_, b_, _ in cases if d_ == depth))]
        circuits[depth] = (circuit, evs)
    circuit, evs = circuits[depth]
    nl, stride = depth, 2 * depth
    ctx = N.Context.get(0)
    dev = GKRProtocol._device_circuit(circuit, ctx)
    ptrs = (C.c_void_p * (B * (nl + 1)))(*[t.data_ptr() for ev in evs[:B] for t in ev])
    lens = (C.c_size_t * (nl + 1))(*[t.shape[0] for t in evs[0]])
    sums, n_rounds = np.zeros((B, nl, 4), np.uint64), np.zeros((B, nl), np.uint32)
    rp_lens, rps = np.zeros((B, nl, stride), np.uint32), np.zeros((B, nl, stride, 7, 2, 4), np.uint64)
    wb, wc, w0 = np.zeros((B, nl, 4), np.uint64), np.zeros((B, nl, 4), np.uint64), np.zeros((B, 2, 4), np.uint64)
    status = np.zeros(B, np.int32)

    def call():
        N.check(N.lib().zkhip_gkr_prove_batch(dev.handle, C.c_uint32(B), C.c_uint32(lanes), ptrs, lens, p(sums), p(n_rounds), p(rp_lens), p(rps), p(wb), p(wc), p(w0),
                                              None, p(status)), "batch")
    call(); call()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    print("depth %2d  B %3d  lanes %d : %7.3f ms per proof (call %.2f ms)" % (depth, B, lanes, 1e3 * sorted(ts)[2] / B, 1e3 * sorted(ts)[2]), flush=True)

"""GKRProtocol.prove from several host threads at once, a context each (zkhip contexts are per thread): a proof keeps ONE workgroup busy
most of the time, so independent proofs share the chip.  usage: python tools/gkr_threads.py [depth] [threads...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zk_cryptography_amd as zk

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 20
counts = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
per_thread = 4


def worker(k, barrier, out):
    with torch.cuda.stream(torch.cuda.Stream()):          # a stream per thread: the context follows torch's current stream, and the
        _worker(k, barrier, out)                          # default stream is one queue for every thread


def _worker(k, barrier, out):
    circuit, ev = CIRCUIT, EV                             # one Circuit for all threads: a device copy per context (thread)
    zk.GKRProtocol.prove(circuit, ev)
    zk.GKRProtocol.prove(circuit, ev)
    torch.cuda.synchronize()
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(per_thread):
        zk.GKRProtocol.prove(circuit, ev)
    out[k] = (t0, time.perf_counter())


CIRCUIT = zk.Circuit.random(depth)
EV = CIRCUIT.evaluation(zk.Fr.synthetic(2 ** depth, 0x5EED000000002001))
torch.cuda.synchronize()
for n in counts:
    barrier = threading.Barrier(n)
    out = [None] * n
    ts = [threading.Thread(target=worker, args=(k, barrier, out)) for k in range(n)]
    for t in ts: t.start()
    for t in ts: t.join()
    span = max(b for _, b in out) - min(a for a, _ in out)
    print("depth %d, %d threads: %.3f ms per proof (each thread alone: %.3f ms)" % (depth, n, 1e3 * span / (n * per_thread),
          1e3 * sum(b - a for a, b in out) / (n * per_thread)))

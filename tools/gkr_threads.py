"""GKRProtocol.prove from several host threads at once, a context and a stream each (zkhip contexts are per thread; the Python mirror's
context follows torch's current stream, and the default stream is ONE queue for every thread): a proof keeps one workgroup busy most
of the time, so independent proofs share the chip.  One Circuit for all threads (the mirror keeps a device copy per context); every
thread's proofs are compared with a synchronous one.
usage: python tools/gkr_threads.py [depth] [threads...]        e.g.  20 1 2 4 8
       python tools/gkr_threads.py --json 8:8 20:4,8           one JSON object {"depth_8": {"8": ms}, ...} (bench.py's leg)"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zk_cryptography_amd as zk

PER_THREAD = 4


def run(depth, counts):
    circuit = zk.Circuit.random(depth)
    ev = circuit.evaluation(zk.Fr.synthetic(2 ** depth, 0x5EED000000002001))
    want = [p.to_bytes() for p in zk.GKRProtocol.prove(circuit, ev).sumcheck_proofs]
    torch.cuda.synchronize()
    res = {}
    for n in counts:
        barrier, out, same = threading.Barrier(n), [None] * n, [False] * n

        def worker(k):
            with torch.cuda.stream(torch.cuda.Stream()):
                zk.GKRProtocol.prove(circuit, ev)
                zk.GKRProtocol.prove(circuit, ev)
                torch.cuda.synchronize()
                barrier.wait()
                t0 = time.perf_counter()
                for _ in range(PER_THREAD):
                    proof = zk.GKRProtocol.prove(circuit, ev)
                out[k] = (t0, time.perf_counter())
                same[k] = [p.to_bytes() for p in proof.sumcheck_proofs] == want
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(n)]
        for t in ts: t.start()
        for t in ts: t.join()
        assert all(same), "a thread's proof differs from the synchronous one"
        span = max(b for _, b in out) - min(a for a, _ in out)
        res[n] = (1e3 * span / (n * PER_THREAD), 1e3 * sum(b - a for a, b in out) / (n * PER_THREAD))
    return res


if len(sys.argv) > 1 and sys.argv[1] == "--json":
    obj = {}
    for spec in sys.argv[2:]:
        d, cs = spec.split(":")
        obj["depth_%s" % d] = {str(n): round(v[0], 3) for n, v in run(int(d), [int(c) for c in cs.split(",")]).items()}
    print(json.dumps(obj))
else:
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for n, (ms, alone) in run(depth, [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]).items():
        print("depth %d, %d threads: %.3f ms per proof (each thread's own calls: %.3f ms)" % (depth, n, ms, alone))

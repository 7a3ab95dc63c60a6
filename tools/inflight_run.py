"""Sumcheck proofs IN FLIGHT, nothing else: `proofs` proofs of 2^log_n entries over 8 tables round robin with up to `depth` in flight (the loop of
bench.py's `pipelined` leg), no library events -- the program to put under `rocprofv3 --kernel-trace` (tools/trace_passes.py reads the trace).
usage: python tools/inflight_run.py [log_n] [depth] [proofs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zk_cryptography_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
proofs = int(sys.argv[3]) if len(sys.argv) > 3 else 48
mixed = os.environ.get("INFLIGHT_MIXED") == "1"        # tables 3 and 7 half as long: which proof a pass belongs to shows in its grid size
polys = [zk.Multilinear(torch.randint(0, 2 ** 62, (1 << (log_n - (1 if mixed and t % 4 == 3 else 0)), 4), dtype=torch.int64, device="cuda")) for t in range(max(8, depth))]


extra = [torch.cuda.Stream() for _ in range(int(os.environ.get("INFLIGHT_EXTRA_STREAMS", "0")))]      # idle streams of the caller's: do they cost the pipeline its hardware queues?
if os.environ.get("INFLIGHT_SYNC_FIRST") == "1":        # a synchronous proof first: the context's own fold stream and side streams come into being before the lanes
    sc = zk.Sumcheck(polys[0]); sc.poly_sum(); sc.prove()


def run(k):
    pend = []
    for i in range(k):
        sc = zk.Sumcheck(polys[i % len(polys)]); sc.poly_sum()
        pend.append(sc.prove_begin())
        if len(pend) == depth:
            pend.pop(0).wait()
    for h in pend:
        h.wait()
    torch.cuda.synchronize()


run(2 * depth)
res = []
for _ in range(5):
    t0 = time.perf_counter(); run(proofs); res.append((time.perf_counter() - t0) / proofs * 1e6)
print("depth %d: %.1f us per proof (median of 5 runs of %d proofs; %s)" % (depth, sorted(res)[2], proofs, " ".join("%.1f" % r for r in res)))

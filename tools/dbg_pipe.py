import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import zk_cryptography_amd as zk
n = 1 << 24
tabs = [torch.from_numpy(zk.Fr.synthetic(n, 0x5EED000000000001 + 8 * i).view(np.int64)).cuda() for i in range(2)]
polys = [zk.Multilinear(t) for t in tabs]
def run(k):
    pend = None
    for i in range(k):
        sc = zk.Sumcheck(polys[i & 1]); sc.poly_sum()
        h = sc.prove_begin()
        if pend is not None:
            pend.wait()
        pend = h
    pend.wait()
    torch.cuda.synchronize()
def sync(k):
    for i in range(k):
        sc = zk.Sumcheck(polys[0]); sc.poly_sum(); sc.prove()
    torch.cuda.synchronize()
sync(5)
t0 = time.perf_counter(); sync(40); print("sync %.1f us" % ((time.perf_counter() - t0) / 40 * 1e6))
run(4)
for k in (20, 40, 40):
    t0 = time.perf_counter(); run(k); print("in flight x%d: %.1f us per proof" % (k, (time.perf_counter() - t0) / k * 1e6))
t0 = time.perf_counter(); sync(40); print("sync %.1f us" % ((time.perf_counter() - t0) / 40 * 1e6))
t0 = time.perf_counter(); run(40); print("in flight x40: %.1f us per proof" % ((time.perf_counter() - t0) / 40 * 1e6))

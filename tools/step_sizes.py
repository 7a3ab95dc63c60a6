import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import zk_cryptography_amd as zk
for lg in [int(a) for a in sys.argv[1:]] or [19, 20, 21, 22, 23]:
    poly = zk.Multilinear(zk.Fr.synthetic_device(1 << lg, 77 + lg) if hasattr(zk.Fr, "synthetic_device") else torch.randint(0, 2**62, ((1 << lg), 4), dtype=torch.int64, device="cuda"))
    def step():
        sc = zk.Sumcheck(poly); sc.poly_sum(); return sc.prove()
    for _ in range(5): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    print("2^%d: %.1f us per poly_sum + prove (median of 30)" % (lg, 1e6 * sorted(ts)[15]))

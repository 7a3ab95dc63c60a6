"""Diagnostic: what the SRS cache guard of zk_cryptography_amd.kzg.TrustedSetup costs per call (it runs before every commitment / opening that
uses a derived table)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import zk_cryptography_amd as zk
tau = zk.Fr.random(16, 5)
srs = zk.TrustedSetup.setup(tau).precompute()
torch.cuda.synchronize()
def timed(name, fn, reps=500):
    fn(); t = time.perf_counter()
    for _ in range(reps): fn()
    print(name, "%.1f us" % ((time.perf_counter() - t) / reps * 1e6), flush=True)
for rep in range(2):
    timed("table property", lambda: srs.table)
    timed("_fingerprint", srs._fingerprint)
    timed("_check_caches", srs._check_caches)
    timed("_stamp", srs._stamp)
poly = zk.Multilinear(zk.Fr.random(1 << 16, 3))
timed("commit 2^16", lambda: zk.MultilinearKZG.commitment(poly, srs), 50)

"""The k-variable fold ALONE (no serial kernel beside it): partial_evaluations(points, [0] * k) on a 2^log_n table, the fold launch timed by
the library's HIP events (zkhip_profile).  usage: [ZKHIP_MF=cfg] python tools/perf_multifold.py [log_n] [k ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zk_cryptography_amd as zk
from zk_cryptography_amd import _native as N

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ks = [int(a) for a in sys.argv[2:]] or [6]
g = torch.Generator(device="cuda").manual_seed(1)
tab = torch.randint(0, 2 ** 62, (1 << log_n, 4), dtype=torch.int64, device="cuda", generator=g)
poly = zk.Multilinear(tab)
ctx = N.Context.get()
for k in ks:
    pts = zk.Fr.random(k, 5)
    for _ in range(3):
        poly.partial_evaluations(pts, [0] * k)
    torch.cuda.synchronize()
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "profile_enable")
    for _ in range(20):
        poly.partial_evaluations(pts, [0] * k)
    ms, cnt, by = C.c_double(), C.c_uint64(), C.c_double()
    N.check(N.lib().zkhip_profile_read(ctx.handle, b"multifold", C.byref(ms), C.byref(cnt), C.byref(by)), "profile_read")
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "profile_enable")
    us = 1e3 * ms.value / max(1, cnt.value)
    print("MF=%s 2^%d k=%d: %.2f us per fold, %.0f GB/s (%.3f of 8 TB/s), %d launches" % (
        os.environ.get("ZKHIP_MF", "default"), log_n, k, us, by.value / cnt.value / us / 1e3, by.value / cnt.value / us / 1e3 / 8000, cnt.value))

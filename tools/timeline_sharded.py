"""The same timeline for the SHARDED prover protocol (zkhip_sc_prove_sharded), world = 1, no transport:
kernel durations and the gaps between them.  usage: python tools/timeline.py [log_n]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk
from zk_cryptography_amd import _native as N
from zk_cryptography_amd import distributed as D
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
t = torch.randint(0, 2 ** 62, (1 << log_n, 4), dtype=torch.int64, device="cuda")
poly = zk.Multilinear(t)
for _ in range(5):
    D.ShardedSumcheck(D.HipSumcheckEngine(t), 1).prove()
ctx = N.Context.get()
N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "enable")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    D.ShardedSumcheck(D.HipSumcheckEngine(t), 1).prove()
wall = (time.perf_counter() - t0) / 3
mx = 256
names = C.create_string_buffer(32 * mx)
st, sp = (C.c_double * mx)(), (C.c_double * mx)()
cnt = C.c_uint32(0)
N.check(N.lib().zkhip_profile_timeline(ctx.handle, mx, names, st, sp, C.byref(cnt)), "timeline")
N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "disable")
rows = sorted((st[i], sp[i], names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode()) for i in range(cnt.value))
prev = None
for a, b, nm in rows:
    print("%9.1f %9.1f  %7.1f us  %-16s %s" % (a, b, b - a, nm, ("gap %5.1f" % (a - prev)) if prev is not None else ""))
    prev = b if prev is None else max(prev, b)
print("wall per step (events attached): %.1f us" % (wall * 1e6))

"""Timeline of Sumcheck proofs IN FLIGHT (two tables alternately, zkhip_sumcheck_prove_begin / _end) from the library's HIP events:
which kernels of the two proofs overlap.  usage: python tools/timeline_pipelined.py [log_n] [proofs]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk
from zk_cryptography_amd import _native as N
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
polys = [zk.Multilinear(torch.randint(0, 2 ** 62, (1 << log_n, 4), dtype=torch.int64, device="cuda")) for _ in range(2)]


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


run(6)
t0 = time.perf_counter(); run(20); print("in flight, no events: %.1f us per proof" % ((time.perf_counter() - t0) / 20 * 1e6))
ctx = N.Context.get()
N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "enable")
t0 = time.perf_counter()
run(proofs)
wall = (time.perf_counter() - t0) / proofs
mx = 512
names = C.create_string_buffer(32 * mx)
st, sp = (C.c_double * mx)(), (C.c_double * mx)()
cnt = C.c_uint32(0)
N.check(N.lib().zkhip_profile_timeline(ctx.handle, mx, names, st, sp, C.byref(cnt)), "timeline")
N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "disable")
rows = sorted((st[i], sp[i], names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode()) for i in range(cnt.value))
for a, b, nm in rows:
    print("%9.1f %9.1f  %7.1f us  %-16s" % (a, b, b - a, nm))
print("wall per proof (events attached): %.1f us" % (wall * 1e6))

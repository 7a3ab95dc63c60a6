"""Stream-ordered timeline (the library's own HIP events: kernel groups and the gaps between them) and wall time of one call of
a chosen entry point.  usage: python tools/timeline_any.py <what> [<what> ...]
  eval24             Multilinear.evaluation at 2^24
  commit8 / commit12 / commit16 ...   MultilinearKZG.commitment on a 2^k-point SRS (plain), commitT8 ... with the shifted table
  open12 ...         MultilinearKZG.open at 2^k
  k5_20 / k5_22 / k3_20 ...   ComposedSumcheck.prove on a product of K tables of 2^n entries
  m23_20             MultiComposedSumcheckProver.prove_partial on (2 + 3) tables of 2^20
  gkr8 / gkr20       GKRProtocol.prove on Circuit.random(depth)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zk_cryptography_amd as zk
from zk_cryptography_amd import _native as N

g = torch.Generator(device="cuda").manual_seed(1)


def rnd(n):
    t = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    return t


def make(what):
    if what.startswith("eval"):
        k = int(what[4:])
        poly, pts = zk.Multilinear(rnd(1 << k)), zk.Fr.random(k, 4)
        return lambda: poly.evaluation(pts)
    if what.startswith("commit"):
        table = what[6] == "T"
        k = int(what[7:] if table else what[6:])
        srs = zk.TrustedSetup.setup(zk.Fr.random(k, 5))
        if table:
            srs.precompute()
        else:
            srs = zk.TrustedSetup(srs.powers_of_tau_in_g1, srs.inf)
        poly = zk.Multilinear(rnd(1 << k))
        return lambda: zk.MultilinearKZG.commitment(poly, srs)
    if what.startswith("open"):
        k = int(what[4:])
        srs = zk.TrustedSetup.setup(zk.Fr.random(k, 5))
        srs = zk.TrustedSetup(srs.powers_of_tau_in_g1, srs.inf)
        poly, z = zk.Multilinear(rnd(1 << k)), zk.Fr.random(k, 6)
        return lambda: zk.MultilinearKZG.open(poly, z, srs)
    if what[0] == "k":
        K, n = int(what[1]), 1 << int(what.split("_")[1])
        sc = zk.ComposedSumcheck(zk.ComposedMultilinear([rnd(n) for _ in range(K)]))
        return lambda: sc.prove()
    if what[0] == "m":
        shape, n = [int(ch) for ch in what[1:].split("_")[0]], 1 << int(what.split("_")[1])
        terms = [zk.ComposedMultilinear([rnd(n) for _ in range(k)]) for k in shape]
        s = zk.MultiComposedSumcheckProver.calculate_poly_sum(terms)
        return lambda: zk.MultiComposedSumcheckProver.prove_partial(terms, s)
    if what.startswith("gkr"):
        d = int(what[3:])
        c = zk.Circuit.random(d)
        ev = c.evaluation(zk.Fr.random(2 ** d, 7))
        return lambda: zk.GKRProtocol.prove(c, ev)
    raise SystemExit("unknown: " + what)


for what in sys.argv[1:]:
    fn = make(what)
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ctx = N.Context.get()
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "enable")
    fn()
    mx = 1024
    names = C.create_string_buffer(32 * mx)
    st, sp = (C.c_double * mx)(), (C.c_double * mx)()
    cnt = C.c_uint32(0)
    N.check(N.lib().zkhip_profile_timeline(ctx.handle, mx, names, st, sp, C.byref(cnt)), "timeline")
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "disable")
    rows = sorted((st[i], sp[i], names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode()) for i in range(cnt.value))
    print("== %s: wall median %.1f us, min %.1f us (10 calls, no events attached); %d scopes" % (what, 1e6 * sorted(ts)[5], 1e6 * min(ts), len(rows)))
    prev = None
    for a, b, nm in rows[:120]:
        print("%9.1f %9.1f  %7.1f us  %-22s %s" % (a, b, b - a, nm, ("gap %5.1f" % (a - prev)) if prev is not None else ""))
        prev = b if prev is None else max(prev, b)
    sys.stdout.flush()

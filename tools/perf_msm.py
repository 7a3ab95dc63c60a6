"""Diagnostic: KZG-commit (MSM) time per size on one GPU + the tiling property at full size.

Property (size independent): for an SRS made of T copies of a base SRS B, commit(tile(B, T), s) must equal
commit(B, sum_t s[t*|B| : (t+1)*|B|]) -- the same group element by linearity, computed through two different
bucket populations (the tiled run sees every base point T times, in T different buckets per window).
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk
from zk_cryptography_amd.field import R_MOD

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 14
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 24
base_log = 16
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"   # uniform | ones | bytes (skewed bucket populations)
one = torch.from_numpy(zk.Fr.from_ints([1]).view(np.int64)).cuda()
g = torch.Generator(device="cuda").manual_seed(7)
tau = zk.Fr.random(base_log, 11)
t0 = time.perf_counter()
base = zk.TrustedSetup.setup(tau)
torch.cuda.synchronize()
print("SRS 2^%d generated in %.1f ms" % (base_log, (time.perf_counter() - t0) * 1e3), flush=True)

for log_n in range(lo, hi + 1):
    n = 1 << log_n
    if log_n <= base_log:
        srs = zk.TrustedSetup(base.powers_of_tau_in_g1[:n], base.inf[:n])
    else:
        T = n >> base_log
        srs = zk.TrustedSetup(base.powers_of_tau_in_g1.repeat(T, 1), base.inf.repeat(T))
    sc = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    if kind == "ones":
        sc = one.repeat(n, 1)
    elif kind == "bytes":   # canonical values < 256: a residue x < 256 times R (mod r) via Multilinear * Fr
        sc[:, 1:] = 0
        sc[:, 0] &= 255
        sc = (zk.Multilinear(sc) * zk.Fr.from_int(pow(2, 256, R_MOD))).evaluations
    poly = zk.Multilinear(sc)
    if os.environ.get("TABLE"):
        t0 = time.perf_counter(); srs.precompute(); torch.cuda.synchronize()
        print("    table built in %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    c = zk.MultilinearKZG.commitment(poly, srs)
    torch.cuda.synchronize()
    reps = 5 if log_n <= 22 else 2
    t = time.perf_counter()
    for _ in range(reps):
        c2 = zk.MultilinearKZG.commitment(poly, srs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    ok = ""
    if log_n > base_log:
        T = n >> base_log
        acc = zk.Multilinear(sc[: 1 << base_log])
        for t_ in range(1, T):
            acc = acc + zk.Multilinear(sc[t_ << base_log: (t_ + 1) << base_log])
        ref = zk.MultilinearKZG.commitment(acc, base)
        ok = "tiling property %s" % ("OK" if ref == c and c2 == c else "MISMATCH")
    print("commit 2^%d: %.3f ms  %.1f M points/s  %s" % (log_n, dt * 1e3, n / dt / 1e6, ok), flush=True)
    if os.environ.get("PHASES"):
        import ctypes as C
        from zk_cryptography_amd import _native as N
        ctx = N.Context.get()
        N.lib().zkhip_profile_enable(ctx.handle, 1)
        zk.MultilinearKZG.commitment(poly, srs)
        out = []
        for name in (b"msm_convert_points", b"msm_sort", b"msm_order", b"msm_accumulate", b"msm_overflow", b"msm_segment", b"msm_terms"):
            ms, cnt, by = C.c_double(0), C.c_uint64(0), C.c_double(0)
            N.lib().zkhip_profile_read(ctx.handle, name, C.byref(ms), C.byref(cnt), C.byref(by))
            out.append("%s %.3f" % (name.decode()[4:], ms.value))
        N.lib().zkhip_profile_enable(ctx.handle, 0)
        print("    phases (ms): " + ", ".join(out), flush=True)
    del srs, poly, sc
    torch.cuda.empty_cache()

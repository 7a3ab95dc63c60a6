"""Diagnostic: MultilinearKZG::open time per size on one GPU (folded SRS cached vs derived per call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk

for log_n in [int(a) for a in sys.argv[1:]] or [12, 16, 20]:
    tau, z = zk.Fr.random(log_n, 5), zk.Fr.random(log_n, 6)
    srs = zk.TrustedSetup.setup(tau)
    g = torch.Generator(device="cuda").manual_seed(3)
    poly = zk.Multilinear(torch.randint(0, 2 ** 62, (1 << log_n, 4), dtype=torch.int64, device="cuda", generator=g))
    modes = {"derived": False, "cached": True, "tables": "tables"}
    for cached in [modes[m] for m in os.environ.get("PERF_OPEN_MODES", "derived,cached,tables").split(",")]:
        if cached == "tables":
            t = time.perf_counter()
            srs.precompute_open()
            torch.cuda.synchronize()
            print("   level tables 2^%d: built in %.1f ms, %.2f GiB" % (log_n, (time.perf_counter() - t) * 1e3, srs.level_tables.numel() / 2 ** 30), flush=True)
        zk.MultilinearKZG.open(poly, z, srs, cache_folded_srs=bool(cached))
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 10 if cached else 3
        for _ in range(reps):
            zk.MultilinearKZG.open(poly, z, srs, cache_folded_srs=bool(cached))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print("open 2^%d (%s folded SRS): %.2f ms" % (log_n, "level tables +" if cached == "tables" else "cached" if cached else "derived per call", dt * 1e3), flush=True)
    t = time.perf_counter()
    zk.MultilinearKZG.commitment(poly, srs); torch.cuda.synchronize()
    print("   one commitment 2^%d: %.2f ms" % (log_n, (time.perf_counter() - t) * 1e3), flush=True)

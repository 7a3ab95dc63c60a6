"""Diagnostic: Sumcheck::poly_sum + prove across table sizes on one GPU (looking for cliffs between the kernel plans)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zk_cryptography_amd as zk

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 28
g = torch.Generator(device="cuda").manual_seed(1)
for log_n in range(lo, hi + 1):
    n = 1 << log_n
    poly = zk.Multilinear(torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g))
    def step():
        s = zk.Sumcheck(poly); s.poly_sum(); return s.prove()
    step(); torch.cuda.synchronize()
    reps = 20 if log_n <= 24 else 5
    t = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("2^%d: %.3f ms  %.2f G field-evals/s  (%.1f us per round)" % (log_n, dt * 1e3, n / dt / 1e9, dt * 1e6 / max(1, log_n)), flush=True)
    del poly
    torch.cuda.empty_cache()

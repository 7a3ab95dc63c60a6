"""Diagnostic timings of the composed / multi-composed sumcheck provers on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zk_cryptography_amd as zk

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda n: torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
logs = [int(a) for a in sys.argv[1:]] or [20]
for log_n in logs:
    n = 1 << log_n
    for k in (2, 3, 5):
        cm = zk.ComposedMultilinear([rnd(n) for _ in range(k)])
        sc = zk.ComposedSumcheck(cm)
        t = timeit(lambda: sc.prove(), 5)
        print("composed prove K=%d 2^%d: %.3f ms  (%.1f us/round, %.2f G evals/s)" % (k, log_n, t * 1e3, t * 1e6 / log_n, k * n / t / 1e9), flush=True)
    for shape in ((2, 2), (2, 3), (3, 3, 2)):
        terms = [zk.ComposedMultilinear([rnd(n) for _ in range(k)]) for k in shape]
        s = zk.MultiComposedSumcheckProver.calculate_poly_sum(terms)
        t = timeit(lambda: zk.MultiComposedSumcheckProver.prove_partial(terms, s), 5)
        print("multi-composed prove_partial %s 2^%d: %.3f ms  (%.1f us/round)" % (shape, log_n, t * 1e3, t * 1e6 / log_n), flush=True)

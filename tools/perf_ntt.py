"""Diagnostic: NTT (Domain::fft / ifft) and UnivariateEval::multiply times per size, with both roofs --
HBM (64 n bytes per pass) and field products (n/2 log2 n butterflies of ~360 VALU instructions)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zk_cryptography_amd as zk

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda n: torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
VALU_PEAK = 1024 * 2.4e9 / 4.9          # wave-instructions per second (tools/ubench.hip)
for log_n in [int(a) for a in sys.argv[1:]] or (12, 16, 18, 20, 21, 22, 24):
    n = 1 << log_n
    x = rnd(n); d = zk.Domain(n)
    for name, fn in (("fft ", lambda: d.fft(x)), ("ifft", lambda: d.ifft(x))):
        t = timeit(fn)
        products = n / 2 * log_n
        print("%s 2^%d: %.3f ms  | %.2f G butterflies/s = %.0f%% of the VALU issue peak at 360 instr each | HBM: %d passes x 64 n = %.0f GB/s"
              % (name, log_n, t * 1e3, products / t / 1e9, 100 * products * 360 / 64 / t / VALU_PEAK,
                 1 + (max(log_n - 8, 0) + 6) // 7, (1 + (max(log_n - 8, 0) + 6) // 7) * 64 * n / t / 1e9), flush=True)
a, b = zk.DenseUnivariatePolynomial(rnd(1 << 20)), zk.DenseUnivariatePolynomial(rnd(1 << 20))
t = timeit(lambda: zk.UnivariateEval.multiply(a, b), 5)
print("multiply 2^20 x 2^20: %.3f ms" % (t * 1e3))

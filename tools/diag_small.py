"""Diagnostic only: where a round of the serial prover kernel spends its time (libzkhip_diag.so, -DZK_STAMPS; s_memtime ticks at the 2.4 GHz core clock, printed in
microseconds).  Rows: one per round (wave 0: hash block 1 / block 2 / publish / barrier wait / product; helpers: when the first and
the last helper wave reach the barrier, relative to wave 0), then one per serial kernel (prologue, rounds)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zk_cryptography_amd import _native as N
N.LIB_PATH = os.path.join(N.CSRC, "libzkhip_diag.so")
import zk_cryptography_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << log_n
t = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda")
poly = zk.Multilinear(t)
for _ in range(3):
    sc = zk.Sumcheck(poly); sc.poly_sum(); sc.prove()
buf = np.zeros((64, 8), dtype=np.uint64)
N.lib().zkhip_debug_read_stamps(N.Context.get().handle, buf.ctypes.data_as(C.c_void_p))
tick = 1.0 / 2400.0   # us per s_memtime tick: the counter runs at the 2.4 GHz core clock of an otherwise idle chip (calibrated: a steady round = 5.4 us)
prev_end = None
for r in range(log_n):
    s = buf[r].astype(np.int64)
    print("round %2d  block1 %5.2f  block2 %5.2f  publish %5.2f  barrier %5.2f  product %5.2f | first helper at barrier %+6.2f, last helper %+6.2f (vs wave 0 arrival)   round total %5.2f us"
          % (r, (s[1] - s[0]) * tick, (s[2] - s[1]) * tick, (s[3] - s[2]) * tick, (s[4] - s[3]) * tick, (s[5] - s[4]) * tick,
             (s[7] - s[3]) * tick, (s[6] - s[3]) * tick, ((s[0] - prev_end) * tick if prev_end else 0.0) + (s[5] - s[0]) * tick))
    prev_end = s[5]
for r0 in range(24):
    s = buf[40 + r0].astype(np.int64)
    if s[0]:
        print("kernel starting at round %2d: prologue %6.2f us, rounds %6.2f us" % (r0, (s[1] - s[0]) * tick, (s[2] - s[1]) * tick))

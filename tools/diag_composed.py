"""Diagnostic only: where a round of the composed / multi-composed prover spends its time (libzkhip_diag.so, -DZK_STAMPS; s_memtime
ticks of the 2.4 GHz core clock printed in microseconds).  Per round: [sums or record reduction] | items (interpolation, canonical
forms) | message | schedules | hash | challenge + publish | fold.   usage: python tools/diag_composed.py [log_n] [multi | k3 | k4 | k5]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zk_cryptography_amd import _native as N
N.LIB_PATH = os.environ.get("ZKHIP_DIAG_LIB") or os.path.join(N.CSRC, "libzkhip_diag.so")
import zk_cryptography_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
multi = len(sys.argv) > 2 and sys.argv[2] == "multi"
kk = int(sys.argv[2][1:]) if len(sys.argv) > 2 and sys.argv[2][0] == "k" else 2
n = 1 << log_n
tabs = [zk.Multilinear(torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda")) for _ in range(max(4, kk))]
if kk > 2:
    cs = zk.ComposedSumcheck(zk.ComposedMultilinear(tabs[:kk]))
    run = lambda: cs.prove()
elif multi:
    poly = [zk.ComposedMultilinear(tabs[:2]), zk.ComposedMultilinear(tabs[2:])]
    claimed = zk.MultiComposedSumcheckProver.calculate_poly_sum(poly)
    run = lambda: zk.MultiComposedSumcheckProver.prove_partial(poly, claimed)
else:
    cs = zk.ComposedSumcheck(zk.ComposedMultilinear(tabs[:2]))
    run = lambda: cs.prove()
for _ in range(3):
    run()
buf = np.zeros((64, 8), dtype=np.uint64)
N.lib().zkhip_debug_read_stamps(N.Context.get().handle, buf.ctypes.data_as(C.c_void_p))
tick = 1.0 / 2400.0
prev = None
for r in range(log_n):
    s = buf[r].astype(np.int64)
    d = lambda a, b: (s[b] - s[a]) * tick
    fold = d(5, 7) if s[7] > s[5] else float("nan")
    print("round %2d  start->close %6.2f  items %5.2f  message %5.2f  schedules %5.2f  hash %5.2f  publish %5.2f  fold %5.2f | since previous round's start %7.2f us"
          % (r, d(6, 0), d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), fold, (s[6] - prev) * tick if prev else 0.0))
    prev = s[6]
print("cross workgroup (the first one), per round it runs beside: wait | tiles | record + signal | closer's round start -> this workgroup's start (us)")
for r in range(min(log_n, 32)):
    s = buf[32 + r].astype(np.int64)
    if s[1] == 0:
        continue
    w = (s[1] - s[0]) * tick if s[0] else float("nan")
    print("round %2d  wait %6.2f  tiles %6.2f  record %6.2f  | start - closer's start %7.2f" % (r, w, (s[2] - s[1]) * tick, (s[3] - s[2]) * tick, (s[1] - buf[r].astype(np.int64)[6]) * tick))

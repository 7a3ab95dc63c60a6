"""Time of one Sumcheck poly_sum + prove per table size (median of 30 calls) -- the figures behind overlapped_min_log (csrc/zkhip.hip):
run once as is and once with ZKHIP_OVERLAP_MIN_LOG=19 to compare the stage and the overlapped plan at every size.
usage (GPU box): python tools/step_sizes.py [log_n ...]      (default: 19 20 21 22 23)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zk_cryptography_amd as zk
for lg in [int(a) for a in sys.argv[1:]] or [19, 20, 21, 22, 23]:
    poly = zk.Multilinear(torch.randint(0, 2**62, ((1 << lg), 4), dtype=torch.int64, device="cuda"))
    def step():
        sc = zk.Sumcheck(poly); sc.poly_sum(); return sc.prove()
    for _ in range(5): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    print("2^%d: %.1f us per poly_sum + prove (median of 30)" % (lg, 1e6 * sorted(ts)[15]))

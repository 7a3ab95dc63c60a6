"""Busy / idle analysis of a rocprofv3 kernel trace (…_kernel_trace.csv): for the last `frac` of the trace, the time covered by
kernels, the idle time between them, the gap histogram and per-kernel totals.  usage: python tools/trace_gaps.py trace.csv [frac]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.33
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_lo = ev[-1][1] - (ev[-1][1] - ev[0][0]) * frac
ev = [e for e in ev if e[0] >= t_lo]
busy, idle, cur_end = 0, 0, ev[0][0]
gaps = []
per = defaultdict(lambda: [0, 0])
for a, b, nm in ev:
    if a > cur_end:
        idle += a - cur_end
        gaps.append((a - cur_end, nm))
        busy += b - a
        cur_end = b
    elif b > cur_end:
        busy += b - cur_end
        cur_end = b
    per[nm.split("(")[0][-60:]][0] += 1
    per[nm.split("(")[0][-60:]][1] += b - a
span = cur_end - ev[0][0]
print("span %.3f ms  busy %.3f ms  idle %.3f ms  kernels %d" % (span / 1e6, busy / 1e6, idle / 1e6, len(ev)))
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 1e4), (1e4, 3e4), (3e4, 1e5), (1e5, 1e9)):
    g = [x for x, _ in gaps if lo <= x < hi]
    print("  gaps %6.0f..%6.0f us: %5d  total %.3f ms" % (lo / 1e3, hi / 1e3, len(g), sum(g) / 1e6))
print("  largest gaps before:", [(round(x / 1e3, 1), n.split("(")[0][-30:]) for x, n in sorted(gaps, reverse=True)[:8]])
for nm, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %-60s %6d  %9.3f ms  avg %7.1f us" % (nm, c, t / 1e6, t / c / 1e3))

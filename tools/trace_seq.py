"""The kernel sequence of the tail of a rocprofv3 kernel trace (…_kernel_trace.csv): start offset, duration, gap before, name --
and the same folded per kernel name.  usage: python tools/trace_seq.py trace.csv [span_ms_from_the_end] [max_rows]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 400
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_hi = ev[-1][1]
ev = [e for e in ev if e[0] >= t_hi - span_ms * 1e6]
t0 = ev[0][0]
prev = None
per = defaultdict(lambda: [0, 0.0, 0.0])
for i, (a, b, nm) in enumerate(ev):
    short = nm.split("(")[0].replace("zk::", "").replace("void ", "")[:44]
    gap = (a - prev) / 1e3 if prev is not None else 0.0
    if i < max_rows:
        print("%9.1f  %7.1f us  gap %6.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, gap, short))
    per[short][0] += 1
    per[short][1] += (b - a) / 1e3
    per[short][2] += max(gap, 0.0)
    prev = b if prev is None else max(prev, b)
print("---- per kernel: calls, busy us, gap-before us (sum)")
for nm, (c, t, g) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-46s %5d  %9.1f  %9.1f" % (nm, c, t, g))
print("span %.1f us" % ((ev[-1][1] - t0) / 1e3))

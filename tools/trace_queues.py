"""Which hardware queue every stream of a rocprofv3 kernel trace ran on, and how busy each queue was.  usage: python tools/trace_queues.py <dir of the trace> [frac of the trace, from the end]"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r.get("Stream_Id", "?")) for r in rows)
t1 = ev[-1][1]; t0 = t1 - (t1 - ev[0][0]) * frac
ev = [e for e in ev if e[0] >= t0]
per = defaultdict(lambda: defaultdict(lambda: [0, 0]))
for a, b, q, s in ev:
    per[q][s][0] += 1; per[q][s][1] += b - a
span = ev[-1][1] - ev[0][0]
for q in sorted(per, key=lambda x: int(x)):
    tot = sum(v[1] for v in per[q].values())
    print("queue %2s: busy %5.1f %%  streams " % (q, 100.0 * tot / span) + "  ".join("s%s (%d kernels, %.1f %%)" % (s, v[0], 100.0 * v[1] / span) for s, v in sorted(per[q].items(), key=lambda kv: int(kv[0]) if kv[0].isdigit() else 0)))

"""How the streaming passes of proofs in flight share the chip, from a rocprofv3 kernel trace of tools/inflight_run.py: over the last `frac` of
the trace, the time during which 0 / 1 / 2 / 3+ streaming passes (fine_sums, multifold) were running, the passes' durations alone and beside
another one, and the wall time per proof.  usage: python tools/trace_passes.py <kernel_trace.csv> [frac]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3


def short(nm):
    for k in ("fine_sums", "multifold", "blockfold", "group_sums", "sumcheck_small"):
        if k in nm:
            return k
    return nm.split("(")[0][-24:]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
t_hi = ev[-1][1]
t_lo = t_hi - (t_hi - ev[0][0]) * frac
ev = [e for e in ev if e[0] >= t_lo]
passes = [e for e in ev if e[2] in ("fine_sums", "multifold")]
n_proofs = sum(1 for e in passes if e[2] == "multifold")
span = ev[-1][1] - ev[0][0]
print("window %.3f ms, %d proofs (multifold launches): %.1f us per proof" % (span / 1e6, n_proofs, span / 1e3 / max(n_proofs, 1)))
pts = []
for a, b, _, _ in passes:
    pts.append((a, 1)); pts.append((b, -1))
pts.sort()
level, last, at = 0, ev[0][0], defaultdict(int)
for t, d in pts:
    at[min(level, 3)] += t - last
    last, level = t, level + d
at[0] += ev[-1][1] - last
for l in range(4):
    print("  %s streaming passes running: %8.3f ms  %5.1f %%" % ("3+" if l == 3 else str(l), at[l] / 1e6, 100.0 * at[l] / span))
# a pass's duration by how much of it was shared with another pass
for kind in ("fine_sums", "multifold"):
    alone, shared = [], []
    for a, b, k, _ in passes:
        if k != kind:
            continue
        ov = sum(max(0, min(b, b2) - max(a, a2)) for a2, b2, _, _ in passes if (a2, b2) != (a, b) and a2 < b and b2 > a)
        (alone if ov < 0.1 * (b - a) else shared).append((b - a) / 1e3)
    for nm, v in (("alone", alone), ("beside another pass", shared)):
        if v:
            print("  %-10s %-20s n=%3d  avg %6.1f us  min %6.1f  max %6.1f" % (kind, nm, len(v), sum(v) / len(v), min(v), max(v)))
per = defaultdict(lambda: [0, 0])
for a, b, k, _ in ev:
    per[k][0] += 1; per[k][1] += b - a
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-24s %5d launches  avg %7.1f us  %6.1f us per proof" % (k, c, t / c / 1e3, t / 1e3 / max(n_proofs, 1)))
queues = defaultdict(int)
for e in ev:
    queues[e[3]] += 1
print("  hardware queues in use:", dict(queues))

"""A slice of a rocprofv3 kernel trace in start order: times (us), duration, hardware queue, stream, dispatch id, grid, kernel.
usage: python tools/trace_slice.py <dir of the trace> [first kernel as a fraction of the trace | 'run' = after the last long idle gap] [kernels]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r.get("Stream_Id", ""), int(r["Dispatch_Id"]), int(r["Grid_Size_X"])) for r in rows)
t0, n = ev[0][0], len(ev)
where = sys.argv[2] if len(sys.argv) > 2 else "0.8"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 70
if where == "run":
    first, end = 0, ev[0][1]
    for i, e in enumerate(ev):
        if e[0] - end > 300e3: first = i          # the host's pause between two runs
        end = max(end, e[1])
else:
    first = int(n * float(where))


def short(nm):
    for k in ("fine_sums", "multifold", "blockfold", "group_sums", "sumcheck_small"):
        if k in nm: return k
    return nm.split("(")[0][-24:]


for a, b, nm, q, s, d, g in ev[first:first + count]:
    print("%10.1f %10.1f %7.1f  q%s s%-3s disp %6d grid %8d %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, s, d, g, short(nm)))

"""prints the kernels of the last `count` launches of `marker` onwards from a rocprofv3 kernel trace CSV: start, end, duration, name"""
import csv, glob, os, sys
d, marker, count = sys.argv[1], sys.argv[2], int(sys.argv[3])
f = sorted(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "zk::" in r["Kernel_Name"]]
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
start = idx[-count]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f %7.1f us %-44s" % (a / 1e3, b / 1e3, (b - a) / 1e3, r["Kernel_Name"].split("(")[0][-44:]))

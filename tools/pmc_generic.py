"""Per-kernel averages of whatever counters a set of rocprofv3 --pmc passes collected.
usage: python tools/pmc_generic.py <counter_collection.csv> [...]"""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()[-40:]
        if "multifold" in name or "fine_sums" in name or "fold_kernel" in name:
            acc["%s g=%s" % (name, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-32s avg %16.2f  (n=%d)" % (c, sum(v) / len(v), len(v)))

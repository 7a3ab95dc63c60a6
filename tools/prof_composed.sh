#!/bin/bash
# rocprofv3 kernel trace of ComposedSumcheck::prove (two tables, 2^22; GPU box, repo root): the last prove's kernels ordered by start time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
rm -rf gpurun_out/r03/composed
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/composed -- python3 tools/prof_composed_k2.py ${1:-22} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03/composed/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last prove: from the last launch of the first kernel name of a prove
names = [r["Kernel_Name"].split("(")[0][-44:] for r in rows]
first = names[-1]
idx = [i for i, nm in enumerate(names) if "cross2" in nm or "round" in nm]
# find the start of the last prove as the last big gap
starts = [i for i in range(1, len(rows)) if int(rows[i]["Start_Timestamp"]) - int(rows[i-1]["End_Timestamp"]) > 30000]
s = starts[-1] if starts else 0
t0 = int(rows[s]["Start_Timestamp"])
prev = t0
for r, nm in zip(rows[s:], names[s:]):
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f %7.1f us gap %5.1f  %-44s grid %s" % (a / 1e3, b / 1e3, (b - a) / 1e3, (int(r["Start_Timestamp"]) - prev) / 1e3, nm, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev = int(r["End_Timestamp"])
PY

#!/bin/bash
# rocprofv3 kernel trace of MultilinearKZG::open at 2^20 (GPU box, repo root): per-kernel stats + the trace ordered by start time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/open -- python3 tools/perf_open.py 20 > gpurun_out/r03/open.txt 2>&1
grep "open 2\|one commit" gpurun_out/r03/open.txt
f=$(ls gpurun_out/r03/open/*/*kernel_stats.csv | head -1); head -22 "$f" | cut -d, -f1-5 | sed 's/(.*),/",/' | cut -c1-150
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r03/open/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last open call: find the last open_step_kernel run of 20 launches
idx = [i for i, r in enumerate(rows) if "open_step" in r["Kernel_Name"]]
start = idx[-20]
t0 = int(rows[start]["Start_Timestamp"])
out = open('gpurun_out/r03/open_last_call_trace.txt', 'w')
for r in rows[start:]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if a > 12e6: break
    out.write("%9.1f %9.1f %8.1f us  q%-3s %-40s grid %s\n" % (a / 1e3, b / 1e3, (b - a) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
out.close()
PY
wc -l gpurun_out/r03/open_last_call_trace.txt


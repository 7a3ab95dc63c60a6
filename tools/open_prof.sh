#!/bin/bash
# per-kernel durations of MultilinearKZG::open at 2^20, plain batch against the level tables (tools/perf_open.py under rocprofv3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in cached tables; do
  PERF_OPEN_MODES=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/open_$m -- python3 $R/tools/perf_open.py 20 > $R/gpurun_out/open_$m.log 2>&1 < /dev/null
  grep "^open\|level" $R/gpurun_out/open_$m.log
  f=$(find $R/gpurun_out/open_$m -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then head -24 "$f" | cut -c1-150; fi
done

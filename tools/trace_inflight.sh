#!/bin/bash
# rocprofv3 kernel trace of tools/inflight_run.py at several depths + tools/trace_passes.py on each.  usage: tools/trace_inflight.sh [log_n] [depths...]
R=$GRAFT_REPO_ROOT; L=${1:-24}; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for D in ${@:-6}; do
  O=$R/gpurun_out/inflight_d$D; rm -rf $O; mkdir -p $O
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/inflight_run.py $L $D 48 > $O.log 2>&1 < /dev/null
  grep "per proof" $O.log | tail -1
  f=$(find $O -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python3 $R/tools/trace_passes.py "$f" 0.3 < /dev/null
done

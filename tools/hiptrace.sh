#!/bin/bash
# HIP API statistics of one leg of bench.py (host-side costs: staged copies, allocations, long launches).  usage: tools/hiptrace.sh <name> <bench.py flags...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; N=$1; shift
timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/hiptrace_$N -- python3 $R/bench.py "$@" > $R/gpurun_out/hiptrace_$N.log 2>&1 < /dev/null
f=$(find $R/gpurun_out/hiptrace_$N -name '*hip_api_stats.csv' | head -1)
[ -n "$f" ] && grep -v "__hip" "$f" | head -16 | cut -d, -f1-7
find $R/gpurun_out/hiptrace_$N -name '*hip_api_trace.csv' -size +20M -delete

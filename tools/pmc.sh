#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <out-name> "<counters>" <script.py> <args...>
# one rocprofv3 --pmc pass (with --kernel-trace only) as CSV under gpurun_out/<out-name>; bounded by `timeout`.
name=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 500 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/$name -- python3 "$@" > gpurun_out/$name.txt 2>&1
ls gpurun_out/$name/*/ | head

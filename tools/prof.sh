#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <out-name> <script.py> <args...>
# rocprofv3 kernel trace + stats of a python script as CSV under gpurun_out/<out-name>; always bounded by `timeout`.
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -- python3 "$@" > gpurun_out/$name.txt 2>&1
grep -v "^W2026\|amdgpu.ids" gpurun_out/$name.txt | cut -c1-1800
f=$(ls gpurun_out/$name/*/*kernel_stats.csv | head -1)
head -${PROF_LINES:-28} "$f" | cut -d, -f1-4 | sed 's/(.*),/",/' | cut -c1-200

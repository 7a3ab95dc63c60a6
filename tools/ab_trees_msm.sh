#!/bin/bash
# usage (GPU box, repo root): tools/ab_trees_msm.sh -- 2^20-point commits (TABLE=1: shifted-SRS table) of this tree and of a copy of an older one
# under _old_tree/: the call's time, then per-kernel rocprofv3 averages of the same script
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in _old_tree . _old_tree .; do
  ( cd $t && echo "== $t" && TABLE=1 timeout 200 python3 tools/perf_msm.py 20 20 2>&1 | grep "commit 2" && timeout 200 python3 tools/perf_msm.py 20 20 2>&1 | grep "commit 2" )
done
for t in _old_tree .; do
  ( cd $t && rm -rf /tmp/ab_$$ && TABLE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$$ -- python3 tools/perf_msm.py 20 20 > /dev/null 2>&1
    echo "== $t (table)"; f=$(ls /tmp/ab_$$/*/*kernel_stats.csv | head -1); grep -i "msm" "$f" | sed 's/(.*)"/"/' | cut -d, -f1-4 | head -20 )
done

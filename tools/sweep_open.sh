#!/bin/bash
# usage (GPU box, repo root): tools/sweep_open.sh -- MultilinearKZG::open at 2^20 per setting of the grouping knobs (ZKHIP_OPEN_*), two runs each
for rep in 1 2; do
  for mid in 17 18 19; do for wf in 0 1; do for bl in 12 13 14; do
    echo -n "MID=$mid WIDE_FIRST=$wf BATCH_LOG=$bl: "
    ZKHIP_OPEN_MID_LOG=$mid ZKHIP_OPEN_WIDE_FIRST=$wf ZKHIP_OPEN_BATCH_LOG=$bl timeout 120 python tools/perf_open.py 20 2>&1 | grep cached
  done; done; done
done

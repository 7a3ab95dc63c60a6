#!/bin/bash
# usage (GPU box, repo root): tools/sweep_open.sh [sizes] -- MultilinearKZG::open per setting of the batch's window-width knob
# (ZKHIP_MSM_BATCH_DELTA: width = log2(n_j) - delta, raised until the sort's partitions fit) and of the pipelines form, two runs each
for rep in 1 2; do
  for d in 1 2 3 4 5; do
    echo "DELTA=$d"
    ZKHIP_MSM_BATCH_DELTA=$d timeout 200 python tools/perf_open.py ${1:-12 16 20 22} 2>&1 | grep cached
  done
  echo "PIPELINES=1"
  ZKHIP_OPEN_PIPELINES=1 timeout 200 python tools/perf_open.py ${1:-12 16 20 22} 2>&1 | grep cached
done

#!/bin/bash
# A/B of the streaming passes' occupancy caps on the single proof and on proofs in flight.  usage: tools/ab_inflight.sh "<ZKHIP_FINE_LDS> <ZKHIP_MF_OCC>" ...
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  export ZKHIP_FINE_LDS=$1 ZKHIP_MF_OCC=$2
  echo "== fine_lds $1 mf_occ $2"
  timeout 100 python3 $R/tools/step_sizes.py 24 2>&1 < /dev/null | tail -1
  timeout 60 python3 $R/tools/timeline.py 24 2>&1 < /dev/null | grep -E "fine_sums|multifold" | head -2
  for d in 4 6 8; do timeout 100 python3 $R/tools/inflight_run.py 24 $d 96 < /dev/null | tail -1; done
done

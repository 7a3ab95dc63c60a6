#!/bin/bash
# same-box A/B of the composed / GKR provers: ZKHIP_PIPE=0 (round by round) against the default (one round ahead of the transcript)
for rep in 1 2; do for v in 0 1; do
  ZKHIP_PIPE=$v python bench.py --no-msm --no-ntt --no-h2d --no-fold --no-cpu-baseline --no-exchange --no-pipelined --steps 5 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ZKHIP_PIPE=$v composed ms_per_prove', d['composed']['ms_per_prove'], 'gkr', d['gkr']['ms_per_proof'])"
done; done

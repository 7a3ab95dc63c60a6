#!/bin/bash
# same-box sweep of the pipelined mid rounds: tile workgroups at most (ZKHIP_PIPE_WGS)
for w in 512 256 128 64; do
  ZKHIP_PIPE_WGS=$w python bench.py --no-msm --no-ntt --no-h2d --no-fold --no-cpu-baseline --no-exchange --no-pipelined --steps 5 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('WGS=$w composed ms_per_prove', d['composed']['ms_per_prove'], 'gkr', d['gkr']['ms_per_proof'])"
done

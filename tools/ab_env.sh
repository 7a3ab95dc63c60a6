#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh VAR "v1 v2" [reps] -- the prover's bench leg per setting of one diagnostic environment variable
for rep in $(seq 1 ${3:-2}); do
  for cfg in $2; do
    env $1=$cfg python bench.py --steps 40 --warmup 5 --no-msm --no-ntt --no-composed --no-gkr --no-cpu-baseline --no-fold --no-h2d --no-exchange 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1=$cfg', 'ms_per_step', d['ms_per_step'], 'batches', d.get('batches'), 'pipelined', (d.get('pipelined') or {}).get('ms_per_step'))"
  done
done

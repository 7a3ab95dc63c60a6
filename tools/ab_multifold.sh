#!/bin/bash
# usage (GPU box, repo root): tools/ab_multifold.sh "0 1 3" -- the prover's bench leg per ZKHIP_MF setting (0 = VALU form, r = rotation of the term order)
for cfg in $1; do
  for rep in 1 2; do
    ZKHIP_MF=$cfg python bench.py --steps 40 --warmup 5 --no-msm --no-ntt --no-composed --no-gkr --no-cpu-baseline --no-fold 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('MF=$cfg', 'ms_per_step', d['ms_per_step'], 'pipelined', (d.get('pipelined') or {}).get('ms_per_step'), 'fold_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
  done
done

#!/bin/bash
# usage (GPU box): tools/ab_bench.sh "<ENV=1 ...>" ... -- one short sumcheck-only bench line per environment setting
for e in "$@"; do
  echo "== $e"
  env $e python bench.py --no-cpu-baseline --no-msm --no-composed --no-gkr --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step', d['ms_per_step'], 'multifold_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])
    elif l: print(l[:300])
"
done

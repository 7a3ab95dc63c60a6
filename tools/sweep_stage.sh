#!/bin/bash
# same-box sweep: from which table size on two rounds per pass (the stage form) beat round-by-round launches -- one-term claims
# (ZKHIP_STAGE_MIN_LOG_ONE: ComposedSumcheck) and claims of several terms (ZKHIP_STAGE_MIN_LOG_MANY: the GKR layers)
run() { python bench.py --no-msm --no-ntt --no-h2d --no-fold --no-cpu-baseline --no-exchange --no-pipelined --steps 5 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1 composed ms_per_prove', d['composed']['ms_per_prove'], 'gkr', d['gkr']['ms_per_proof'])"; }
for l in 21 20 19 18 17; do ZKHIP_STAGE_MIN_LOG_ONE=$l run "ONE=$l"; done
for l in 18 17 16 15 14; do ZKHIP_STAGE_MIN_LOG_MANY=$l run "MANY=$l"; done

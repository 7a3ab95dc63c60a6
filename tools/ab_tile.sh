#!/bin/bash
# same-box A/B of a compile-time choice of the composed provers (here: ZK_PIPE_TILE, the tile of the pipelined mid rounds): the shipped
# library against a second build.  Build the second one first, in zk-cryptography_amd/csrc:
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DZK_PIPE_TILE=32 -c -o build/composed_alt.o composed.hip
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o libzkhip_alt.so build/zkhip.o build/composed_alt.o build/msm.o build/ntt.o build/gkr.o build/shard.o
for rep in 1 2 3; do for l in libzkhip_alt.so libzkhip.so; do
  ZKHIP_LIB=zk-cryptography_amd/csrc/$l python tools/bench_with_lib.py --no-msm --no-ntt --no-h2d --no-fold --no-cpu-baseline --no-exchange --no-pipelined --steps 5 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$l composed ms_per_prove', d['composed']['ms_per_prove'], 'gkr', d['gkr']['ms_per_proof'])"
done; done

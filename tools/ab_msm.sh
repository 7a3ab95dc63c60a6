for rep in 1 2 3; do for l in libzkhip_alt.so libzkhip.so; do
  ZKHIP_LIB=zk-cryptography_amd/csrc/$l python tools/bench_with_lib.py --steps 10 --warmup 2 --no-cpu-baseline --no-composed --no-gkr --no-ntt --no-h2d --no-exchange --no-pipelined --no-fold 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d['msm']
print('$l commit', m['ms_per_commit'], 'in flight', m['pipelined']['ms_per_commit'], 'plain', m['without_srs_table']['ms_per_commit'], 'open', m['extras']['open']['ms_per_open'], m['extras']['open_level_tables']['ms_per_open'])"
done; done

timeout 600 python -m pytest tests/test_gpu_kzg.py tests/test_cpp_mirror.py tests/test_gpu_threads.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
run() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-composed --no-gkr --no-ntt --no-h2d --no-exchange --no-pipelined --no-fold 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); m=d['msm']
print('$1 commit', m['ms_per_commit'], 'in flight', m['pipelined']['ms_per_commit'], 'plain', m['without_srs_table']['ms_per_commit'], 'open', m['extras']['open']['ms_per_open'], m['extras']['open_level_tables']['ms_per_open'])"; }
run default; ZKHIP_LEVEL_TABLE_DELTA=1 run delta1; run default
python tools/perf_fingerprint.py 2>&1 | grep commit

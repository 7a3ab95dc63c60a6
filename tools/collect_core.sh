#!/bin/bash
# usage (GPU box, repo root): tools/collect_core.sh <round>      -- the part of tools/collect.sh the bench line is checked against:
# rocprofv3 kernel stats of the default bench and of the prover leg, the PMC FETCH / WRITE passes of the prover leg (-> pmc_traffic.json with
# the kernel-source fingerprint bench.py's roofline.traffic_age compares), the library's own timelines, the default bench line, the depth-20
# GKR proof's kernel stats and the batch leg's figures.  Every step bounded by `timeout`, nothing reads stdin.
R=${1:?round name, e.g. r06}
mkdir -p gpurun_out/$R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
exec < /dev/null
P="--steps 3 --warmup 1 --no-cpu-baseline --no-msm --no-composed --no-gkr --no-ntt --no-h2d --no-exchange --no-pipelined --no-gkr-threads"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python3 bench.py --no-cpu-baseline --no-gkr-threads > gpurun_out/$R/bench_under_rocprof.json 2> gpurun_out/$R/stats.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats_prover -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-msm --no-composed --no-gkr --no-ntt --no-h2d --no-exchange --no-pipelined --no-gkr-threads > gpurun_out/$R/bench_prover_under_rocprof.json 2> gpurun_out/$R/stats_prover.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_fetch -- python3 bench.py $P > /dev/null 2> gpurun_out/$R/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_write -- python3 bench.py $P > /dev/null 2> gpurun_out/$R/pmc_write.err
timeout 200 python3 tools/timeline.py 24 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -30 > gpurun_out/$R/b_prover_timeline.txt
timeout 200 python3 tools/timeline_pipelined.py 24 8 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/$R/c_proofs_in_flight_timeline.txt
# proofs in flight: the kernel trace of six in flight (which pass ran when, on which hardware queue), and the figure at every depth without a profiler
{
  echo "rocprofv3 --kernel-trace of tools/inflight_run.py 24 6 48 (2^24 entries, six proofs in flight, 8 tables round robin): tools/trace_passes.py over the"
  echo "last 30 % of the trace, then tools/trace_slice.py (start, end, duration in us; hardware queue; stream; dispatch id; grid; kernel).  s0 = the caller's"
  echo "stream: the sums pass of table i + 1, the big fold of proof i - 3, ... back to back; the other streams = the four serial streams of the lanes."
  rm -rf gpurun_out/$R/inflight; mkdir -p gpurun_out/$R/inflight
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/inflight -- python3 tools/inflight_run.py 24 6 48 2>&1 | grep "per proof"
  f=$(find gpurun_out/$R/inflight -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python3 tools/trace_passes.py "$f" 0.3 && echo && python3 tools/trace_slice.py gpurun_out/$R/inflight 0.75 44
  echo; echo "without the profiler (median of five runs of 96 proofs):"
  for d in 2 3 4 6 8; do timeout 100 python3 tools/inflight_run.py 24 $d 96 2>&1 | grep "per proof"; done
} > gpurun_out/$R/c_proofs_in_flight_trace.txt 2>&1
timeout 400 python3 bench.py > gpurun_out/$R/a_bench_line_default_run.json 2> gpurun_out/$R/a_bench_line_default_run.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/gkr20 -- python3 tools/gkr_run.py 20 > gpurun_out/$R/gkr20.txt 2>&1
timeout 200 python3 tools/perf_gkr_batch.py 8:8:8 8:32:8 8:48:12 20:8:8 20:24:12 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$R/e_gkr_batch_ms_per_proof.txt
timeout 300 python3 tools/timeline_any.py k5_22 k5_20 m23_20 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$R/d_k5_timelines.txt
find gpurun_out/$R -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out/$R -name "*.db" -delete
du -sh gpurun_out/$R

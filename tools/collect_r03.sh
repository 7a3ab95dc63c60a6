#!/bin/bash
# Round-3 profile collection on the GPU box (repo root): rocprofv3 kernel stats of the default bench, PMC FETCH / WRITE passes of the
# prover leg, an SQ pass over the NTT leg.  Every profiler run is bounded by `timeout`; outputs under gpurun_out/r03/.
mkdir -p gpurun_out/r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P="--steps 3 --warmup 1 --no-cpu-baseline --no-msm --no-composed --no-gkr --no-ntt --no-h2d --no-exchange --no-pipelined"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/stats -- python3 bench.py --no-cpu-baseline > gpurun_out/r03/bench_under_rocprof.json 2> gpurun_out/r03/stats.err
# the prover leg alone (no proofs in flight, whose overlapped kernels run longer): the averages the roofline object is checked against
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/stats_prover -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-msm --no-composed --no-gkr --no-ntt --no-h2d --no-exchange --no-pipelined > gpurun_out/r03/bench_prover_under_rocprof.json 2> gpurun_out/r03/stats_prover.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_fetch -- python3 bench.py $P > /dev/null 2> gpurun_out/r03/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_write -- python3 bench.py $P > /dev/null 2> gpurun_out/r03/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_sq_ntt -- python3 tools/perf_ntt.py 21 > gpurun_out/r03/pmc_sq_ntt.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_sq_prover -- python3 bench.py $P > /dev/null 2> gpurun_out/r03/pmc_sq_prover.err
timeout 200 python3 tools/timeline.py 24 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -30 > gpurun_out/r03/b_prover_timeline.txt
timeout 200 python3 tools/timeline_pipelined.py 24 8 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/r03/c_proofs_in_flight_timeline.txt
timeout 300 python3 bench.py > gpurun_out/r03/a_bench_line_default_run.json 2> gpurun_out/r03/a_bench_line_default_run.err
find gpurun_out/r03 -name "*.csv" | head -20
f=$(ls gpurun_out/r03/stats/*/*kernel_stats.csv | head -1); head -30 "$f" | cut -d, -f1-5 | sed 's/(.*),/",/' | cut -c1-160
# keep what is small: stats CSVs and counter collections (the kernel traces themselves are large)
find gpurun_out/r03 -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out/r03 -name "*.db" -delete
du -sh gpurun_out/r03

#!/bin/bash
# usage (GPU box, repo root): tools/collect.sh <round, e.g. r04>
# A round's profile collection: rocprofv3 kernel stats of the default bench and of the prover leg, PMC FETCH / WRITE passes of the
# prover leg, SQ passes, the library's own timelines, the composed / GKR provers' kernel trace and in-kernel round stamps, the same-box
# A/B of the pipelined rounds, the read-stream probes.  Every profiler run is bounded by `timeout`; outputs under gpurun_out/<round>/.
R=${1:?round name, e.g. r04}
mkdir -p gpurun_out/$R
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (--no-gkr-threads in every profiled run: that leg's child process would inherit the profiler's preload and its 4 / 8-thread proofs would
# land in the kernel averages the roofline object is checked against)
P="--steps 3 --warmup 1 --no-cpu-baseline --no-msm --no-composed --no-gkr --no-ntt --no-h2d --no-exchange --no-pipelined --no-gkr-threads"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python3 bench.py --no-cpu-baseline --no-gkr-threads > gpurun_out/$R/bench_under_rocprof.json 2> gpurun_out/$R/stats.err
# the prover leg alone (no proofs in flight, whose overlapped kernels run longer): the averages the roofline object is checked against
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats_prover -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-msm --no-composed --no-gkr --no-ntt --no-h2d --no-exchange --no-pipelined --no-gkr-threads > gpurun_out/$R/bench_prover_under_rocprof.json 2> gpurun_out/$R/stats_prover.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_fetch -- python3 bench.py $P > /dev/null 2> gpurun_out/$R/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_write -- python3 bench.py $P > /dev/null 2> gpurun_out/$R/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_sq_ntt -- python3 tools/perf_ntt.py 21 > gpurun_out/$R/pmc_sq_ntt.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_sq_prover -- python3 bench.py $P > /dev/null 2> gpurun_out/$R/pmc_sq_prover.err
timeout 200 python3 tools/timeline.py 24 2>&1 | grep -v "^W2026\|amdgpu.ids" | tail -30 > gpurun_out/$R/b_prover_timeline.txt
timeout 200 python3 tools/timeline_pipelined.py 24 8 2>&1 | grep -v "^W2026\|amdgpu.ids" > gpurun_out/$R/c_proofs_in_flight_timeline.txt
timeout 300 python3 bench.py > gpurun_out/$R/a_bench_line_default_run.json 2> gpurun_out/$R/a_bench_line_default_run.err
# the composed / GKR provers: kernel stats of three depth-20 proofs, busy / idle analysis of the trace, in-kernel stamps of a 2 x 2-table
# claim at 2^16 (pipelined launches, then the pipelined tail) and of a K = 2 ComposedSumcheck at 2^22, same-box A/B against ZKHIP_PIPE=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/gkr20 -- python3 tools/gkr_run.py 20 > gpurun_out/$R/gkr20.txt 2>&1
f=$(ls gpurun_out/$R/gkr20/*/*kernel_trace.csv | head -1); python3 tools/trace_gaps.py $f 0.33 > gpurun_out/$R/d_gkr20_trace_gaps.txt 2>&1
make -s -C zk-cryptography_amd/csrc libzkhip_diag.so > /dev/null 2>&1
(echo "== MultiComposedSumcheckProver::prove_partial, 2 terms of 2 tables, 2^16 entries: per-round in-kernel stamps (us)"; timeout 120 python3 tools/diag_composed.py 16 multi 2>&1 | grep "^round"; echo "== ComposedSumcheck::prove, 2 tables, 2^22 entries"; timeout 120 python3 tools/diag_composed.py 22 2>&1 | grep "^round") > gpurun_out/$R/d_composed_round_stamps.txt
timeout 300 ./tools/ab_pipe.sh > gpurun_out/$R/d_ab_pipe.txt 2>&1
# round 5: evaluation as one pass, terms of 3-5 tables, short commits, the GKR outer transcript on the device
timeout 100 python3 tools/timeline_any.py eval24 eval20 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$R/b_evaluation_timeline.txt
timeout 300 python3 tools/timeline_any.py k5_22 k5_20 m23_20 k3_20 k4_20 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$R/d_k5_timelines.txt
(echo "== ComposedSumcheck::prove, 5 tables, 2^20 entries: per-round in-kernel stamps (us)"; timeout 120 python3 tools/diag_composed.py 20 k5 2>&1 | grep "^round") > gpurun_out/$R/d_k5_round_stamps.txt
(timeout 200 python3 tools/timeline_any.py commitT8 commitT10 commitT12 2>&1 | grep -v "amdgpu.ids"; echo "== ZKHIP_MSM_SMALL=0: the bucket pipeline on the same inputs"; ZKHIP_MSM_SMALL=0 timeout 200 python3 tools/timeline_any.py commitT8 commitT12 2>&1 | grep -v "amdgpu.ids") > gpurun_out/$R/e_small_commit_timelines.txt
(for rep in 1 2; do for d in 8 20; do echo "depth $d, outer transcript on the device: $(timeout 100 python3 tools/gkr_run.py $d 2>&1 | grep 'ms per')"; echo "depth $d, outer transcript on the host:   $(ZKHIP_GKR_HOST_TRANSCRIPT=1 timeout 100 python3 tools/gkr_run.py $d 2>&1 | grep 'ms per')"; done; done) > gpurun_out/$R/d_ab_gkr_transcript.txt
f=$(ls gpurun_out/$R/gkr20/*/*kernel_trace.csv | head -1); python3 tools/trace_seq.py $f 9.3 0 > gpurun_out/$R/d_gkr20_kernel_sequence.txt 2>&1
timeout 400 bash tools/sweep_stage.sh > gpurun_out/$R/d_sweep_stage.txt 2>&1
timeout 100 python3 tools/perf_fingerprint.py 2>&1 | grep " us" > gpurun_out/$R/e_srs_guard_cost.txt
[ -x tools/ubench_salu ] && timeout 120 ./tools/ubench_salu > gpurun_out/$R/ubench_salu_gfx950.txt 2>&1
[ -x tools/ubench_sha_split ] && timeout 120 ./tools/ubench_sha_split > gpurun_out/$R/ubench_sha_split_gfx950.txt 2>&1
(for rep in 1 2; do for d in 8 20; do echo "depth $d, small launches fused:   $(timeout 100 python3 tools/gkr_run.py $d 2>&1 | grep 'ms per')"; echo "depth $d, small launches separate: $(ZKHIP_GKR_FUSE_SMALL=0 timeout 100 python3 tools/gkr_run.py $d 2>&1 | grep 'ms per')"; done; done) > gpurun_out/$R/d_ab_gkr_fused_small.txt
(timeout 120 python3 tools/diag_small.py 24 2>&1 | grep "^round\|^kernel") > gpurun_out/$R/b_sumcheck_small_round_stamps.txt
[ -x tools/ubench_fine ] && timeout 120 ./tools/ubench_fine > gpurun_out/$R/ubench_fine_gfx950.txt 2>&1
[ -x tools/ubench_batched_affine ] && timeout 120 ./tools/ubench_batched_affine > gpurun_out/$R/ubench_batched_affine_gfx950.txt 2>&1
# MultilinearKZG::open: plain batch against the level tables -- per-kernel stats at 2^20, time by size, time by window width
for m in cached tables; do
  PERF_OPEN_MODES=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/open_$m -- python3 tools/perf_open.py 20 > gpurun_out/$R/open_$m.log 2>&1 < /dev/null
done
(timeout 300 python3 tools/perf_open.py 12 16 18 20 21 22 2>&1 | grep "open\|level\|commitment"; timeout 300 bash tools/sweep_level_delta.sh 16 18 19 2>&1) > gpurun_out/$R/e_open_by_size_and_width.txt
find gpurun_out/$R -name "*.csv" | head -20
f=$(ls gpurun_out/$R/stats/*/*kernel_stats.csv | head -1); head -30 "$f" | cut -d, -f1-5 | sed 's/(.*),/",/' | cut -c1-160
# keep what is small: stats CSVs and counter collections (the kernel traces themselves are large)
find gpurun_out/$R -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out/$R -name "*.db" -delete
du -sh gpurun_out/$R

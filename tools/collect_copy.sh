#!/bin/bash
# usage (this container, repo root, after `gpurun -- tools/collect.sh <round>` merged gpurun_out/<round>/): tools/collect_copy.sh <round>
# copies what is judged into profiles/<round>/
set -e
N=${1:?round name, e.g. r04}
R=gpurun_out/$N P=profiles/$N
mkdir -p $P
cp $R/stats/runc/*kernel_stats.csv $P/a_bench_kernel_stats.csv
cp $R/bench_under_rocprof.json $P/a_bench_line_under_rocprof.json
cp $R/stats_prover/runc/*kernel_stats.csv $P/a2_prover_kernel_stats.csv
cp $R/bench_prover_under_rocprof.json $P/a2_prover_line_under_rocprof.json
cp $R/a_bench_line_default_run.json $P/
cp $R/b_prover_timeline.txt $R/c_proofs_in_flight_timeline.txt $P/
cp $R/gkr20/runc/*kernel_stats.csv $P/d_gkr20_kernel_stats.csv
cp $R/d_gkr20_trace_gaps.txt $R/d_composed_round_stamps.txt $R/d_ab_pipe.txt $P/
[ -f $R/d_sweep_stage.txt ] && cp $R/d_sweep_stage.txt $P/
for f in b_evaluation_timeline.txt d_k5_timelines.txt d_k5_round_stamps.txt e_small_commit_timelines.txt d_ab_gkr_transcript.txt d_gkr20_kernel_sequence.txt; do [ -f $R/$f ] && cp $R/$f $P/; done
[ -f $R/e_srs_guard_cost.txt ] && cp $R/e_srs_guard_cost.txt $P/
for f in ubench_salu_gfx950.txt ubench_sha_split_gfx950.txt d_ab_gkr_fused_small.txt b_sumcheck_small_round_stamps.txt; do [ -f $R/$f ] && cp $R/$f $P/; done
[ -f $R/ubench_fine_gfx950.txt ] && cp $R/ubench_fine_gfx950.txt $P/
[ -f $R/ubench_batched_affine_gfx950.txt ] && cp $R/ubench_batched_affine_gfx950.txt $P/
[ -f $R/e_open_by_size_and_width.txt ] && cp $R/e_open_by_size_and_width.txt $P/
for m in cached tables; do f=$(find $R/open_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/e_open_${m}_kernel_stats.csv; done
cp $R/pmc_fetch/runc/*counter_collection.csv $P/pmc_fetch_counter_collection.csv
cp $R/pmc_write/runc/*counter_collection.csv $P/pmc_write_counter_collection.csv
python tools/pmc_summary.py $P/pmc_fetch_counter_collection.csv $P/pmc_write_counter_collection.csv $P/pmc_traffic.json | grep -i "multifold\|fine_sums\|fold_kernel"
python tools/bench_summary.py < $P/a_bench_line_default_run.json | head -8
grep "fine_sums\|multifold_mfma\|fold_kernel<false>" $P/a2_prover_kernel_stats.csv | sed 's/(.*)"/"/' | cut -d, -f1-4
grep "fine_sums\|multifold_mfma" $P/a_bench_kernel_stats.csv | sed 's/(.*)"/"/' | cut -d, -f1-4

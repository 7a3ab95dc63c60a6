"""bench.py's multi-rank code path (torchrun, one process per rank, sharded prover / commit / composed prover, max-over-ranks
timing, rank 0 prints the JSON line) exercised on the test box's single GPU: ZKHIP_BENCH_ONE_GPU=1 puts every rank on GPU 0 and
carries the exchange over gloo.  A dry run of the protocol and of the output contract, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_dry_run(world):
    env = dict(os.environ, ZKHIP_BENCH_ONE_GPU="1", ZKHIP_SELFTEST_FAILURE_INJECTION="1")   # + the self-test's opt-in case: one rank fails, nobody hangs
    port = 29800 + (os.getpid() % 1000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--log-n", "22", "--msm-log-n", "12", "--composed-log-n", "15", "--config4-log-n", "12", "--no-cpu-baseline"]   # the gkr leg runs too (replicas)
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only, one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["unit"] == "field-evals/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["launches"] > 0
    assert d["msm"]["value"] > 0
    assert d["composed"]["transcript_replicated_on_all_ranks"] is True and d["composed"]["rounds"] == 15 + (world.bit_length() - 1)
    assert "dry_run" in d
    assert d["config"]["exchanges_per_prove"] >= 1 and d["config"]["transcript_replicated_on_all_ranks"] is True
    assert d["msm"]["commitment_replicated_on_all_ranks"] is True and d["msm"]["exchanges_per_commit"] == 1
    mg = d["multi_gpu"]                                       # strong scaling of the headline + the configs[4]-shaped commit
    assert mg["sumcheck_strong"]["scaling"] == "strong" and mg["sumcheck_strong"]["value"] > 0 and mg["sumcheck_strong"]["exchanges_per_prove"] >= 1
    assert mg["commit_config4_shape"]["commitment_replicated_on_all_ranks"] is True and mg["commit_config4_shape"]["value"] > 0
    assert mg["commit_config4_shape"]["commit_equals_p_tau_times_G"] is True        # ONE SRS sharded over the ranks, identity checked before the timing
    assert list(d)[0] == "legs" and d["legs"]["step_ms"] == d["ms_per_step"] and d["legs"]["config4_commit_ms"] > 0 and d["legs"]["gkr20_ms"] > 0
    assert d["exchange"]["transport"] in ("staged", "rccl", "callback") and "fallback_reason" in d["exchange"] and "rccl_version" in d["exchange"]
    assert d["batches"]["n"] >= 3 and d["batches"]["min"] <= d["ms_per_step"] <= d["batches"]["max"]
    sh = d["gkr"]["sharded"]
    assert sh["proof_equals_single_gpu_proof"] is True and sh["proof_replicated_on_all_ranks"] is True and sh["exchanges_per_proof"] > 0
    # the self-test ran before anything was timed and printed the measured exchange first (stderr: stdout stays ONE line)
    st = [l for l in out.stderr.decode().splitlines() if l.startswith("bench.py selftest: ")]
    assert len(st) == 1
    st = json.loads(st[0][len("bench.py selftest: "):])
    assert st["n_gpus"] == world and all(st["sharded_provers_match_single_gpu_and_rank_0"].values()) and len(st["sharded_provers_match_single_gpu_and_rank_0"]) == 5
    assert st["sharded_provers_match_single_gpu_and_rank_0"]["failure_injection"] is True
    assert st["exchange"]["64_B"]["back_to_back_us"] > 0 and d["exchange"]["64_B"]["back_to_back_us"] > 0

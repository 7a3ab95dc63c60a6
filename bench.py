#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X proving hot path.

Metric (BASELINE.json): field-evals/s of the sumcheck prover over a 24-variable multilinear
(2^24 BLS12-381 Fr evaluations) per GPU.  A "step" is one full pass of the hot path over one
table: Sumcheck::poly_sum + Sumcheck::prove (sumcheck/benches/sumcheck_benchmark.rs:13-22 minus
the verifier) -- fused half-sums + fold kernels with the Fiat-Shamir transcript on the device --
with the table already resident in HBM.  value = (tables' entries consumed by all ranks) / time.

Extra objects on the JSON line: `roofline` for the dominant kernel (the fused fold) from HIP
events on the launch stream, and `cpu_baseline`: the CPU oracle (a C port of the reference
algorithm, single-threaded like the reference) timed on rank 0 at N=1 (plus the same port on all host cores at once).
Informational objects that never enter `value`: `msm` (the second half of BASELINE's metric: KZG commit points/s with its own
roofline and CPU baseline), `composed` (ComposedSumcheck::prove over sharded tables) and `gkr` (GKRProtocol::prove, replicas).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


class _HostStagedCollectives:
    """torch.distributed stand-in for the ZKHIP_BENCH_ONE_GPU dry run: the same calls, CUDA tensors staged through the host."""

    def __init__(self, dist, torch):
        self._d, self._torch = dist, torch
        self.ReduceOp = dist.ReduceOp

    def all_gather_into_tensor(self, out, inp, group=None):
        o = out.cpu()
        self._d.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)

    def all_reduce(self, t, op=None):
        h = t.cpu()
        self._d.all_reduce(h, op=op)
        t.copy_(h)

    def barrier(self):
        self._torch.cuda.synchronize()
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()


_CHILD = """
import sys, time
sys.path.insert(0, %r)
from oracle import oracle as ora
ev = ora.random_fr(1 << 22, 7)
ora.sumcheck_prove(ev)
t = time.perf_counter()
for _ in range(%d):
    ora.sumcheck_prove(ev)
print(time.perf_counter() - t)
"""


def cpu_all_cores(reps=2):
    """The C oracle's poly_sum + prove on all host cores at once: one child process per core (children never touch the GPU)."""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    kids = [subprocess.Popen([sys.executable, "-c", _CHILD % (ROOT, reps)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            for _ in range(cores)]
    times = []
    for k in kids:
        out, _ = k.communicate(timeout=240)
        if k.returncode != 0:
            raise RuntimeError("oracle child exited with %d" % k.returncode)
        times.append(float(out.decode().strip().splitlines()[-1]))
    slowest = max(times)
    return {"value": round(cores * reps * (1 << 22) / slowest, 1), "unit": "field-evals/s", "cores": cores, "kind": "port",
            "sample": "%d processes x %d runs of the same C port on a 2^22-entry table each, slowest process %.1f s (%.1f s with start-up)"
                      % (cores, reps, slowest, time.perf_counter() - t0)}


def bench_msm(args, zk, N, rank, world, barrier, dist, torch, np):
    """KZG commit (MultilinearKZG::commitment) on a 2^msm_log_n-point SRS per GPU, SRS + scalars resident."""
    import ctypes as C
    log_n = args.msm_log_n
    n = 1 << log_n
    tau = zk.Fr.random(log_n, 0x7A0 + rank)
    srs = zk.TrustedSetup.setup(tau)                  # real SRS, generated on the device (not timed)
    plain_srs = zk.TrustedSetup(srs.powers_of_tau_in_g1, srs.inf)
    t_tab = time.perf_counter()
    srs.precompute()                                  # shifted-SRS table (depends on the SRS only; built once, not timed)
    torch.cuda.synchronize()
    t_tab = time.perf_counter() - t_tab
    g = torch.Generator(device="cuda").manual_seed(0x5EED1001 + rank)
    poly = zk.Multilinear(torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g))
    steps = max(2, min(args.steps, 10))
    from zk_cryptography_amd import distributed as D

    def commit():
        if world == 1:
            return zk.MultilinearKZG.commitment(poly, srs)
        # N > 1: this rank's (scalars, SRS) are one shard of a world * 2^log_n commit; partial commitments
        # (104 B per rank) are all-gathered and summed on every rank
        def local():
            c = zk.MultilinearKZG.commitment(poly, srs)
            return c.xy, c.infinity
        return D.sharded_commit(local, D.hip_sum_affine, world, None, dist, device="cuda")

    for _ in range(2):
        com = commit()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        com = commit()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # the same commitments without the table (16 instead of 13 bucket additions per point, 16 bucket reductions)
    zk.MultilinearKZG.commitment(poly, plain_srs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        com_plain = zk.MultilinearKZG.commitment(poly, plain_srs)
    torch.cuda.synchronize()
    dt_plain = time.perf_counter() - t1
    assert world > 1 or com_plain == com, "table and plain commitments differ"
    ctx = N.Context.get()
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "profile_enable")
    zk.MultilinearKZG.commitment(poly, srs)
    ms, cnt, by = C.c_double(), C.c_uint64(), C.c_double()
    N.check(N.lib().zkhip_profile_read(ctx.handle, b"msm_accumulate", C.byref(ms), C.byref(cnt), C.byref(by)), "profile_read")
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "profile_enable")
    out = {"metric": "MSM points/s (KZG commit, 2^%d-point SRS per GPU)" % log_n,
           "value": round(float(n) * world * steps / dt, 1), "unit": "points/s", "ms_per_commit": round(1e3 * dt / steps, 3),
           "steps": steps,
           "srs_table": {"bytes": int(srs._table.numel()), "build_ms": round(1e3 * t_tab, 1),
                         "note": "2^(20 w) * point for the 13 windows of a scalar; depends on the SRS only, built once, not timed"},
           "without_srs_table": {"value": round(float(n) * steps / dt_plain, 1), "unit": "points/s (this rank)",
                                 "ms_per_commit": round(1e3 * dt_plain / steps, 3)},
           "roofline": {"bound": "integer ALU (not HBM: ~10 Fq products of ~900 instructions per bucket addition)",
                        "kernel": "msm_accumulate_kernel", "achieved": round(by.value / (ms.value * 1e-3) / 1e9, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by.value / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "avg_launch_us": round(1e3 * ms.value, 1),
                        "algorithmic_bytes_per_launch": "128 B x points (96 B affine point + 32 B scalar)"}}
    # what actually bounds that kernel: issue of v_mad_u64_u32.  One bucket addition = 10 products in the 14 x 28-bit
    # representation = 4060 multiply-adds; a launch adds one point per non-zero digit (13 windows x points with the
    # shifted-SRS table, all but 2^-20 of them).  Peak = 1024 SIMDs x 64 lanes / 4.9 cycles per wave-instruction x 2.4 GHz as measured by
    # tools/ubench.hip (profiles/r01/ubench_alu_gfx950.txt).
    mads = 4060.0 * 13.0 * n * (1.0 - 2.0 ** -20)
    peak_tmads = 1024 * 64 / 4.9 * 2.4e9 / 1e12
    out["roofline_alu"] = {"bound": "valu", "kernel": "msm_accumulate_kernel", "achieved": round(mads / (ms.value * 1e-3) / 1e12, 2),
                           "peak": round(peak_tmads, 2), "unit": "T v_mad_u64_u32 lane-ops/s",
                           "frac": round(mads / (ms.value * 1e-3) / 1e12 / peak_tmads, 4),
                           "ops_per_launch": "4060 multiply-adds x 13 windows x points"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as ora
        m = 1 << 15                                                     # bounded sample of the naive reference algorithm (~7 s)
        pts = srs.powers_of_tau_in_g1[:m].cpu().numpy().view(np.uint64)
        inf = srs.inf[:m].cpu().numpy()
        jac = np.zeros((m, 18), dtype=np.uint64)
        one = ora.fq_from_ints([1])[0]
        jac[:, :12] = pts
        jac[:, 12:] = np.where(inf[:, None] == 0, one[None, :], 0)
        sc = poly.evaluations[:m].cpu().numpy().view(np.uint64)
        t1 = time.perf_counter()
        ora.kzg_commitment(sc, jac, True)
        cdt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(m / cdt, 1), "unit": "points/s", "cores": 1, "kind": "port",
                               "sample": "C oracle's naive sum of mul_bigint (multilinear_kzg.rs:43-47) on the first 2^15 "
                                         "points/scalars of the same input, %.1f s (cost is linear in points)" % cdt}
    return out


def bench_composed(args, zk, N, rank, world, barrier, dist, torch, np):
    """ComposedSumcheck::prove (composed_sumcheck.rs:32-67; the reference's bench shape: a product of two tables) on
    2^composed_log_n entries per table per GPU.  N > 1: the tables are sharded by low index bits, one record of
    partial sums (96 B) all-gathered per round (zk_cryptography_amd.distributed.ShardedComposedSumcheck)."""
    from zk_cryptography_amd import distributed as D
    K, n = 2, 1 << args.composed_log_n
    g = torch.Generator(device="cuda").manual_seed(0x5EED3001 + rank)
    tables = [torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g) for _ in range(K)]
    poly = zk.ComposedMultilinear([zk.Multilinear(t) for t in tables])

    def prove():
        if world == 1:
            proof, ch = zk.ComposedSumcheck(poly).prove()
            return proof.round_polys, ch
        return D.ShardedComposedSumcheck(D.HipComposedEngine([tables], world, multi=False), world, None, dist).prove()

    steps = max(2, min(args.steps, 10))
    for _ in range(2):
        rp, ch = prove()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        rp, ch = prove()
    barrier()
    dt = time.perf_counter() - t0
    same = True
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mine = torch.from_numpy(np.ascontiguousarray(ch[-1]).view(np.int64)).cuda()   # the last challenge depends on every round
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
    return {"workload": "ComposedSumcheck::prove, product of %d tables, 2^%d entries per table per GPU" % (K, args.composed_log_n),
            "value": round(K * n * world * steps / dt, 1), "unit": "field-evals/s (table entries consumed)",
            "ms_per_prove": round(1e3 * dt / steps, 4), "rounds": int(len(ch)), "steps": steps,
            "transcript_replicated_on_all_ranks": same,
            "sharding": "tables sharded by low index bits, one 96-byte record per rank all-gathered per round" if world > 1 else "single GPU"}


def bench_gkr(args, zk, rank, world, barrier, dist, torch, np):
    """GKRProtocol::prove (gkr/src/protocol.rs:21-117) on Circuit::random(depth) -- the reference's gkr bench shape at depth 8,
    BASELINE configs[3]'s width 2^20 at depth 20.  Replicas only: every rank proves its own circuit."""
    out = {"workload": "GKRProtocol::prove on Circuit::random(depth), evaluation resident in HBM, circuit resident (zkhip_circuit)",
           "replicas": world, "ms_per_proof": {}}
    for depth in (8, 20):
        circuit = zk.Circuit.random(depth)
        ev = circuit.evaluation(zk.Fr.random(2 ** depth, 0x2001 + rank))
        zk.GKRProtocol.prove(circuit, ev)
        barrier()
        reps = 5 if depth <= 8 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            zk.GKRProtocol.prove(circuit, ev)
        barrier()
        dt = (time.perf_counter() - t0) / reps
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        out["ms_per_proof"]["depth_%d" % depth] = round(1e3 * dt, 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=24, help="log2 of the per-GPU table size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--msm-log-n", type=int, default=20, help="log2 of the per-GPU SRS size of the KZG commit leg")
    ap.add_argument("--no-msm", action="store_true")
    ap.add_argument("--composed-log-n", type=int, default=22, help="log2 of the per-GPU table size of the composed-sumcheck leg")
    ap.add_argument("--no-composed", action="store_true")
    ap.add_argument("--no-gkr", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="diagnostic: run the sharded prover protocol even on one GPU")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # diagnostic: ZKHIP_BENCH_ONE_GPU=1 runs the N > 1 code path with every rank on GPU 0 and the exchange over gloo (RCCL
    # refuses two ranks on one device) -- a dry run of the multi-rank protocol on a 1-GPU box, never a measurement
    one_gpu = world > 1 and os.environ.get("ZKHIP_BENCH_ONE_GPU") == "1"
    torch.cuda.set_device(0 if one_gpu else local_rank)
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo")
            dist = _HostStagedCollectives(dist, torch)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import zk_cryptography_amd as zk
    from zk_cryptography_amd import _native as N

    n = 1 << args.log_n
    g = torch.Generator(device="cuda").manual_seed(0x5EED + rank)
    # synthetic random field elements: four limbs each < 2^62, i.e. uniform residues below 2^254 < r
    table = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    poly = zk.Multilinear(table)

    from zk_cryptography_amd import distributed as D

    def step():
        if world == 1 and not args.force_sharded:
            sc = zk.Sumcheck(poly)
            sc.poly_sum()
            return sc.prove()
        # N > 1: ONE prover over the world * 2^log_n-entry table whose rank-interleaved shard is `table`
        # (stage form: one all-gather of 2^k partial block sums per k rounds over RCCL/xGMI + replicated transcript)
        return D.ShardedSumcheck(D.HipSumcheckEngine(table), world, None, dist).prove()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel: HIP events around every fold launch on the launch stream
    ctx = N.Context.get()
    import ctypes as C
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "profile_enable")
    prof_steps = max(1, min(args.steps, 5))
    for _ in range(prof_steps):
        step()
    ms, cnt, by = C.c_double(), C.c_uint64(), C.c_double()
    N.check(N.lib().zkhip_profile_read(ctx.handle, b"multifold", C.byref(ms), C.byref(cnt), C.byref(by)), "profile_read")
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "profile_enable")
    achieved = by.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0
    # HBM bytes of that launch from the PMC counters (separate rocprofv3 --pmc passes; profiles/r01/pmc_traffic.json)
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))
        if args.log_n == 24:
            traffic = pmc["kernels"]["zk::multifold_kernel grid=262144"]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": "multifold_kernel<64> (8-variable fold of the 2^24 table + block sums of its output)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "launches": int(cnt.value), "avg_launch_us": round(1e3 * ms.value / max(1, cnt.value), 2),
                "algorithmic_bytes_per_launch": "32 B x (table entries read + folded entries written), k variables per launch"}

    # ---- second half of BASELINE's metric: MSM points/s of the KZG commit on a 2^20-point SRS per GPU
    msm = None
    if not args.no_msm:
        msm = bench_msm(args, zk, N, rank, world, barrier, dist, torch, np)

    # ---- the composed prover (GKR's sumcheck shape) on sharded tables; informational, never part of `value`
    composed = None
    if not args.no_composed:
        try:
            composed = bench_composed(args, zk, N, rank, world, barrier, dist, torch, np)
        except Exception as e:   # reported, not hidden: the headline legs above are already measured
            composed = {"error": "%s: %s" % (type(e).__name__, e)}

    gkr = None
    if not args.no_gkr:
        try:
            gkr = bench_gkr(args, zk, rank, world, barrier, dist, torch, np)
        except Exception as e:
            gkr = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- CPU baseline: the oracle's single-threaded restatement of poly_sum + prove, rank 0 only.  Last: the all-cores leg
    # loads every host core, which would disturb the host-side share of the GPU legs above if it ran before them
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as ora
        cpu_log = min(args.log_n, 24)
        ev = table[: 1 << cpu_log].cpu().numpy().view(np.uint64)
        reps = 8 if cpu_log >= 24 else 8 << min(6, 24 - cpu_log)
        t1 = time.perf_counter()
        for _ in range(reps):
            ora.sumcheck_prove(ev)
        cdt = time.perf_counter() - t1
        cpu = {"value": round(reps * (1 << cpu_log) / cdt, 1), "unit": "field-evals/s", "cores": 1, "kind": "port",
               "sample": "%d runs of the C oracle's Sumcheck poly_sum+prove (2 Montgomery muls per fold output, as "
                         "evaluation_form.rs:133; single-threaded like the reference) on the same 2^%d-entry table, "
                         "%.1f s in total" % (reps, cpu_log, cdt)}

        # the same port on every host core at once (the reference is single-threaded; this is the box's CPU ceiling for
        # independent provers): one child process per core, each proving its own 2^22-entry table
        try:
            cpu["all_cores"] = cpu_all_cores()
        except Exception as e:
            cpu["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        total_evals = float(n) * world * args.steps
        out = {
            "metric": "field-evals/s (sumcheck 2^%d) + MSM points/s (KZG 2^%d) per BASELINE.json; value = the sumcheck prover's "
                      "field-evals/s, the MSM half is under \"msm\"" % (args.log_n, args.msm_log_n),
            "value": round(total_evals / dt, 1),
            "unit": "field-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            **({"dry_run": "ZKHIP_BENCH_ONE_GPU: all ranks on one GPU, exchange over gloo -- not a measurement"} if one_gpu else {}),
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"field": "BLS12-381 Fr (255-bit Montgomery, 8 x u32 limbs); G1 over Fq (381-bit)",
                       "workload": "24-var multilinear sumcheck prover (poly_sum + prove), BLS12-381 Fr" if args.log_n == 24
                       else "%d-var multilinear sumcheck prover" % args.log_n,
                       "evals_per_gpu": n, "sharding": ("one %d-entry table sharded by low index bits over %d GPUs; per stage of k rounds one RCCL all-gather of the 2^k partial block sums, local k-variable fold, replicated transcript" % (n * world, world))
                       if world > 1 else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "msm": msm,
            "composed": composed,
            "gkr": gkr,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X proving hot path.

Metric (BASELINE.json): field-evals/s of the sumcheck prover over a 24-variable multilinear
(2^24 BLS12-381 Fr evaluations) per GPU.  A "step" is one full pass of the hot path over one
table: Sumcheck::poly_sum + Sumcheck::prove (sumcheck/benches/sumcheck_benchmark.rs:13-22 minus
the verifier) -- block sums, k-variable folds and the Fiat-Shamir transcript on the device --
with the table already resident in HBM.  value = (tables' entries consumed by all ranks) / time.
Timing: after `warmup` steps, batches of EXACTLY `steps` steps, each between barrier + synchronize (MAX over ranks per batch), until
>= 0.3 s are measured; the line reports the median batch and min / median / max under `batches`.

Extra objects on the JSON line: `roofline` for the dominant kernel (the streaming k-variable fold, limb products on the matrix cores)
from HIP events on its launch stream, and `cpu_baseline`: the CPU oracle (a C port of the reference
algorithm, single-threaded like the reference) timed on rank 0 at N=1 (plus the same port on all host cores at once).
Informational objects that never enter `value`: `pipelined` (the same proofs with up to eight in flight), `fold` (the single-variable
fold of SURVEY 8d's 48 n row, with its own roofline), `msm` (the second half of BASELINE's metric: KZG commit points/s with its roofline
and CPU baselines), `ntt` (the 2^21-point transform and the 2^20 x 2^20 product), `composed` (ComposedSumcheck::prove over sharded
tables), `gkr` (GKRProtocol::prove, replicas; at N > 1 also ONE proof sharded), `h2d_inclusive` (the step with the table uploaded over
PCIe first), `exchange` (what one all-gather of the sharded provers costs on this backend), `prediction_n8` (N = 1: the sharded paths
priced for 8 GPUs from their one-rank cost and the exchange cost) and `multi_gpu` (N > 1: strong scaling of the
headline and BASELINE configs[4]'s commit shape).

Synthetic inputs follow SURVEY 8d: uniform field elements from splitmix64-seeded xoshiro256** streams
(zkhip_synthetic_fr): table t of rank g seed 0x5EED000000000001 + t + 16 g, commit scalars 0x5EED000000001001 + g, GKR inputs
0x5EED000000002001 + g.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
VALU_PEAK = 1024 * 2.4e9 / 4.9   # wave-instructions per second: 1024 SIMDs, one v_mad_u64_u32 per 4.9 cycles (tools/ubench.hip)
SEED_TABLE, SEED_SCALARS, SEED_GKR = 0x5EED000000000001, 0x5EED000000001001, 0x5EED000000002001


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

    def broadcast(self, t, src=0):
        h = t.cpu()
        self._d.broadcast(h, src=src)
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

_CHILD_MSM = """
import sys, time
import numpy as np
sys.path.insert(0, %r)
from oracle import oracle as ora
m = %d
g = ora.g1_generator()
jac = np.tile(g, (m, 1))                       # the naive commit's cost does not depend on which points it multiplies
sc = ora.random_fr(m, 11)
t = time.perf_counter()
ora.kzg_commitment(sc, jac, True)
print(time.perf_counter() - t)
"""


MIN_LEG_SECONDS = 0.3    # every leg repeats its batch until this much time has been measured


def _stats(xs, scale=1.0, digits=4):
    xs = sorted(xs)
    return {"n": len(xs), "min": round(scale * xs[0], digits), "median": round(scale * xs[len(xs) // 2], digits), "max": round(scale * xs[-1], digits)}


def _timed(fn, torch, reps=5, min_total=MIN_LEG_SECONDS, max_batches=40):
    """fn() `reps` times per batch, batches until min_total seconds are measured: seconds per call of every batch (rank-local legs)"""
    fn()
    torch.cuda.synchronize()
    out, total = [], 0.0
    while (total < min_total and len(out) < max_batches) or len(out) < 3:
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append(dt / reps)
        total += dt
    return out


def _batches(step, steps, barrier, dist, world, torch, min_total=MIN_LEG_SECONDS, max_batches=60):
    """Batches of EXACTLY `steps` steps, each bracketed by barrier + synchronize, MAX over ranks per batch; as many batches as it takes
    to measure min_total seconds (the same count on every rank: decided on rank 0's clock).  -> (seconds per batch, last result)"""
    def one():
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, r
    dt, res = one()
    n_more = max(2, min(max_batches - 1, int(min_total / max(dt, 1e-9))))
    if world > 1:
        nn = torch.tensor([n_more], dtype=torch.int64, device="cuda")
        dist.broadcast(nn, src=0)
        n_more = int(nn.item())
    out = [dt]
    for _ in range(n_more):
        dt, res = one()
        out.append(dt)
    return out, res


def bench_exchange(comm, world):
    """What ONE exchange of the sharded provers costs, measured INSIDE the library on the communicator the provers use
    (zkhip_comm_measure): all-gathers of 64 B / 8 KiB / 64 KiB per rank issued on the context's stream back to back -- the protocols'
    pattern: an exchange is followed by more enqueues, never by a host wait -- and, as the upper bound, with the host waiting for each.
    At world 1 over a one-rank RCCL communicator this is the fixed cost of the call path that every rank count pays."""
    out = {"transport": comm.transport if isinstance(comm.transport, str) else "callback", "world": world,
           "measured": "inside libzkhip: ncclAllGather on the context's stream + the one-workgroup look at every rank's record for a failed rank's "
                       "poison mark behind it (zkhip_comm_measure: an exchange as the protocols issue it)"}
    if getattr(comm, "fallback_reason", None):      # the library's RCCL communicator did not come up on every rank: host-staged all-gathers over torch.distributed
        out["measured"] = "inside libzkhip over the STAGED transport (device -> host -> torch.distributed all_gather -> device)"
        out["rccl_fallback_reason"] = str(comm.fallback_reason)[:300]
    for nbytes in (64, 8192, 65536):
        b2b, waited = comm.measure(nbytes, 200)
        out["%d_B" % nbytes] = {"back_to_back_us": round(b2b, 2), "with_host_wait_us": round(waited, 2)}
    return out


def selftest(zk, np, torch, N, D, comm, rank, world, dist, same_on_all_ranks):
    """Before anything is timed at N > 1: one small sharded sumcheck / composed / GKR proof and one sharded commit through the in-library
    protocols, every rank's result compared with rank 0's and with the single-GPU prover on the whole input.  Any mismatch ends the job
    with a non-zero status (nothing is swallowed at N > 1)."""
    res = {}
    log_n = 18
    full = zk.Fr.synthetic(1 << log_n, SEED_TABLE + 0x700)                # the same whole table on every rank
    sc = zk.Sumcheck(zk.Multilinear(full))
    sc.poly_sum()
    want, want_ch = sc.prove()
    shard = torch.from_numpy(np.ascontiguousarray(full[rank::world]).view(np.int64)).cuda()
    sh = D.ShardedSumcheck(D.HipSumcheckEngine(shard), world, comm=comm)
    s_, rp_, ch_ = sh.prove()
    ok = np.array_equal(s_, want.sum) and np.array_equal(rp_, want.univariate_poly) and np.array_equal(ch_, want_ch)
    res["sumcheck_2^%d" % log_n] = bool(ok and same_on_all_ranks(np.concatenate([rp_.reshape(-1), ch_.reshape(-1)])))
    tabs = [zk.Fr.synthetic(1 << 16, SEED_TABLE + 0x710 + k) for k in range(2)]
    wproof, wch = zk.ComposedSumcheck(zk.ComposedMultilinear([zk.Multilinear(t) for t in tabs])).prove()
    eng = D.HipComposedEngine([[torch.from_numpy(np.ascontiguousarray(t[rank::world]).view(np.int64)).cuda() for t in tabs]], world, multi=False)
    rp_, ch_ = D.ShardedComposedSumcheck(eng, world, comm=comm).prove()
    ok = np.array_equal(rp_, wproof.round_polys) and np.array_equal(ch_, wch)
    res["composed_2x2^16"] = bool(ok and same_on_all_ranks(np.concatenate([np.asarray(rp_).reshape(-1), ch_.reshape(-1)])))
    circuit = zk.Circuit.random(12)
    ev = circuit.evaluation(zk.Fr.synthetic(1 << 12, SEED_GKR + 0x700))
    wgkr = zk.GKRProtocol.prove(circuit, ev)
    got = zk.GKRProtocol.prove_sharded(circuit, ev, world, rank, comm=comm)
    ok = all(a.to_bytes() == b.to_bytes() for a, b in zip(got.sumcheck_proofs, wgkr.sumcheck_proofs))
    ok = ok and all(np.array_equal(a, b) for a, b in zip(got.wb_s + got.wc_s, wgkr.wb_s + wgkr.wc_s))
    res["gkr_depth_12"] = bool(ok and same_on_all_ranks(np.concatenate([np.asarray(w, dtype=np.uint64).reshape(-1) for w in got.wb_s + got.wc_s])))
    srs = zk.TrustedSetup.setup(zk.Fr.synthetic(12, SEED_SCALARS + 0x700))
    scal = zk.Fr.synthetic(1 << 12, SEED_SCALARS + 0x701)
    wc = zk.MultilinearKZG.commitment(zk.Multilinear(scal), srs)
    xy, inf = D.sharded_commit(srs.powers_of_tau_in_g1[rank::world].contiguous(), srs.inf[rank::world].contiguous(),
                               torch.from_numpy(np.ascontiguousarray(scal[rank::world]).view(np.int64)).cuda(), comm)
    res["commit_2^12"] = bool((not inf) and np.array_equal(xy, wc.xy) and same_on_all_ranks(np.asarray(xy, dtype=np.uint64)))
    if os.environ.get("ZKHIP_SELFTEST_FAILURE_INJECTION", "0") not in ("", "0") and world > 1:
        # opt-in (it has run over gloo and with threads as ranks, never over a real multi-rank RCCL communicator): the last rank fails in front
        # of the second exchange of a sharded sumcheck -- it must get its own status, every other rank ZKHIP_ERR_PEER, nobody may hang, and
        # the next proof on the same communicator must be right again
        if rank == world - 1:
            comm.inject_failure(1, N.ERR_NOMEM)
        try:
            D.ShardedSumcheck(D.HipSumcheckEngine(shard), world, comm=comm).prove()
            outcome = "no status"
        except N.ZkhipPeerError:
            outcome = "peer"
        except N.ZkhipError as e:
            outcome = "own %d" % e.status
        s_, rp_, ch_ = D.ShardedSumcheck(D.HipSumcheckEngine(shard), world, comm=comm).prove()
        res["failure_injection"] = bool(outcome == ("own %d" % N.ERR_NOMEM if rank == world - 1 else "peer") and np.array_equal(ch_, want_ch))
    return res


def _all_cores(child_src, work_per_child, unit, what):
    """One oracle child process per host core at once (children never touch the GPU); value = total work / slowest child."""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    kids = [subprocess.Popen([sys.executable, "-c", child_src], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(cores)]
    times = []
    for k in kids:
        out, _ = k.communicate(timeout=300)
        if k.returncode != 0:
            raise RuntimeError("oracle child exited with %d" % k.returncode)
        times.append(float(out.decode().strip().splitlines()[-1]))
    slowest = max(times)
    return {"value": round(cores * work_per_child / slowest, 1), "unit": unit, "cores": cores, "kind": "port",
            "sample": "%d processes x %s, slowest process %.1f s (%.1f s with start-up)" % (cores, what, slowest, time.perf_counter() - t0)}


def _profile(N, ctx, name):
    import ctypes as C
    ms, cnt, by = C.c_double(), C.c_uint64(), C.c_double()
    N.check(N.lib().zkhip_profile_read(ctx.handle, name, C.byref(ms), C.byref(cnt), C.byref(by)), "profile_read")
    return ms.value, int(cnt.value), by.value


def _synthetic(zk, torch, n, seed):
    import numpy as np
    return torch.from_numpy(zk.Fr.synthetic(n, seed).view(np.int64)).cuda()


def _same_on_all_ranks(dist, torch, np, arr):
    """every rank holds the bytes rank 0 holds?"""
    mine = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1).view(np.int64).copy()).cuda()
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    ok = torch.tensor([1 if torch.equal(ref, mine) else 0], dtype=torch.int64, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return bool(ok.item())


def bench_fold(args, zk, N, poly, torch):
    """MultilinearTrait::partial_evaluation (evaluation_form.rs:123-141) of the 2^log_n table at variable 0: SURVEY 8d's 48 n row."""
    n = len(poly)
    r = zk.Fr.synthetic(1, SEED_TABLE + 0x777)[0]
    ctx = N.Context.get()
    reps = max(3, min(args.steps, 10))
    ts = _timed(lambda: poly.partial_evaluation(r, 0), torch, reps=reps)
    wall = sorted(ts)[len(ts) // 2]
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "profile_enable")
    for _ in range(reps):
        poly.partial_evaluation(r, 0)
    ms, cnt, by = _profile(N, ctx, b"fold")
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "profile_enable")
    ach = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"workload": "partial_evaluation of the 2^%d table at variable 0 (one product per output)" % args.log_n,
            "value": round(n / wall, 1), "unit": "field-evals/s", "ms_per_fold": round(1e3 * wall, 4), "batches": _stats(ts, 1e3),
            "roofline": {"bound": "hbm", "kernel": "fold_kernel<false>", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "launches": cnt, "avg_launch_us": round(1e3 * ms / max(1, cnt), 2),
                         "algorithmic_bytes_per_launch": "48 B x table entries (32 n read + 16 n written)"}}


def bench_evaluate(args, zk, N, poly, torch):
    """MultilinearTrait::evaluation (evaluation_form.rs:162-175) of the 2^log_n table at log_n known points: SURVEY 8d's 96 n row
    (the reference: n folds of a cloned table).  Here ONE pass over the table -- the k-variable fold whose tiles keep the sum of their
    outputs weighted by the eq table of the remaining points -- so the bytes really moved are 32 n, reported beside the
    algorithmic 96 n."""
    n = len(poly)
    pts = zk.Fr.synthetic(args.log_n, SEED_TABLE + 0x778)
    ctx = N.Context.get()
    reps = max(3, min(args.steps, 10))
    ts = _timed(lambda: poly.evaluation(pts), torch, reps=reps)
    wall = sorted(ts)[len(ts) // 2]
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "profile_enable")
    for _ in range(reps):
        poly.evaluation(pts)
    ms, cnt, by = _profile(N, ctx, b"multifold_eval")
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "profile_enable")
    moved = by / max(1, cnt)
    ach = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = None
    try:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "pmc_traffic.json")))
        if files and args.log_n == 24:
            traffic = json.load(open(files[-1])).get("multifold_eval", {}).get("hbm_bytes_per_launch")
    except Exception:
        traffic = None
    return {"workload": "evaluation of the 2^%d table at %d points (host call: the value comes back)" % (args.log_n, args.log_n),
            "value": round(n / wall, 1), "unit": "field-evals/s", "ms_per_evaluation": round(1e3 * wall, 4), "batches": _stats(ts, 1e3),
            "algorithmic_bytes": 96.0 * n, "algorithmic_gbs": round(96.0 * n / wall / 1e9, 1), "algorithmic_frac_of_hbm": round(96.0 * n / wall / 1e9 / HBM_PEAK_GBS, 4),
            "bytes_moved": moved, "moved_frac_of_hbm": round(moved / wall / 1e9 / HBM_PEAK_GBS, 4) if moved else None,
            "roofline": {"bound": "hbm", "kernel": "multifold_mfma_kernel<4, 4, true> (the pass; weights and the records' sum are two small launches beside it)",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "launches": cnt,
                         "avg_launch_us": round(1e3 * ms / max(1, cnt), 2), "traffic": traffic,
                         "algorithmic_bytes_per_launch": "32 B x table entries read"}}


def bench_ntt(args, zk, N, torch):
    """Domain::fft / ifft (domain.rs:108-118) at 2^21 points -- the transform size of a 2^20 x 2^20 product -- and
    UnivariateEval::multiply (evaluation.rs:59-86).  Bound by field products, not HBM: both figures are reported."""
    log_n = args.ntt_log_n
    n = 1 << log_n
    x = _synthetic(zk, torch, n, SEED_TABLE + 0x100)
    d = zk.Domain(n)
    ctx = N.Context.get()
    out = {"workload": "Domain::fft / ifft at 2^%d points; UnivariateEval::multiply 2^%d x 2^%d coefficients" % (log_n, log_n - 1, log_n - 1)}
    reps = max(3, min(args.steps, 10))
    for name, fn in (("fft", lambda: d.fft(x)), ("ifft", lambda: d.ifft(x))):
        ts = _timed(fn, torch, reps=reps)
        out["ms_per_" + name] = round(1e3 * sorted(ts)[len(ts) // 2], 4)
        out["batches_" + name] = _stats(ts, 1e3)
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "profile_enable")
    for _ in range(reps):
        d.fft(x)
    ms1, c1, b1 = _profile(N, ctx, b"ntt_first8")
    ms2, c2, b2 = _profile(N, ctx, b"ntt_pass")
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "profile_enable")
    kernel_ms = (ms1 + ms2) / max(1, c1)
    ach = (b1 + b2) / ((ms1 + ms2) * 1e-3) / 1e9 if ms1 + ms2 > 0 else 0.0
    butterflies = n / 2 * log_n
    valu = butterflies * 380 / 64 / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    out["value"] = round(n / (out["ms_per_fft"] * 1e-3), 1)
    out["unit"] = "points/s (forward transform)"
    out["roofline"] = {"bound": "hbm", "kernel": "ntt_first8_kernel + ntt_pass_kernel (%d passes)" % (1 + c2 // max(1, c1)),
                       "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                       "kernel_us_per_transform": round(1e3 * kernel_ms, 2),
                       "algorithmic_bytes_per_launch": "64 B x points per pass (read + write) + 32 B per twiddle of the pass's table"}
    out["roofline_alu"] = {"bound": "valu", "achieved": round(valu / 1e9, 1), "peak": round(VALU_PEAK / 1e9, 1),
                           "unit": "G wave-instructions/s", "frac": round(valu / VALU_PEAK, 4),
                           "ops_per_launch": "n/2 log2 n butterflies x 380 VALU instructions (one Montgomery product + add + sub; counted in the ISA)"}
    a = zk.DenseUnivariatePolynomial(_synthetic(zk, torch, n // 2, SEED_TABLE + 0x101))
    b = zk.DenseUnivariatePolynomial(_synthetic(zk, torch, n // 2, SEED_TABLE + 0x102))
    ts = _timed(lambda: zk.UnivariateEval.multiply(a, b), torch, reps=reps)
    out["ms_per_multiply"] = round(1e3 * sorted(ts)[len(ts) // 2], 4)
    out["batches_multiply"] = _stats(ts, 1e3)
    return out


def bench_msm(args, zk, N, rank, world, barrier, dist, torch, np):
    """KZG commit (MultilinearKZG::commitment) on a 2^msm_log_n-point SRS per GPU, SRS + scalars resident."""
    log_n = args.msm_log_n
    n = 1 << log_n
    tau = zk.Fr.synthetic(log_n, SEED_SCALARS + 0x100 + rank)
    srs = zk.TrustedSetup.setup(tau)                  # real SRS, generated on the device (not timed)
    plain_srs = zk.TrustedSetup(srs.powers_of_tau_in_g1, srs.inf)
    t_tab = time.perf_counter()
    srs.precompute()                                  # shifted-SRS table (depends on the SRS only; built once, not timed)
    torch.cuda.synchronize()
    t_tab = time.perf_counter() - t_tab
    poly = zk.Multilinear(_synthetic(zk, torch, n, SEED_SCALARS + rank))
    steps = max(2, min(args.steps, 10))
    from zk_cryptography_amd import distributed as D

    def commit():
        if world == 1:
            return zk.MultilinearKZG.commitment(poly, srs)
        # N > 1: this rank's (scalars, SRS) are one shard of a world * 2^log_n commit; the partial commitments (128 B per rank) are
        # all-gathered on the device and summed on every rank, inside zkhip_kzg_commit_sharded
        return D.sharded_commit(None, srs.inf, poly.evaluations, D.Comm.get(N.Context.get(), world, rank, dist), table=srs.table)

    for _ in range(2):
        com = commit()
    commit_batches, com = _batches(commit, steps, barrier, dist, world, torch)
    dt = sorted(commit_batches)[len(commit_batches) // 2]
    same = True
    if world > 1:
        same = _same_on_all_ranks(dist, torch, np, np.concatenate([np.asarray(com[0], dtype=np.uint64), np.array([1 if com[1] else 0], dtype=np.uint64)]))
        assert same, "rank %d: the sharded commitment differs from rank 0's" % rank
    # the same commitments three in flight (zkhip_kzg_commit_begin / _end): the latency-bound reductions and the host epilogue
    # of one commit hide behind the bucket accumulation of the next -- what a prover with several polynomials to commit sees
    pipelined = None
    if world == 1:
        depth = 3                            # zkhip_kzg_commit_begin takes three (measured 2 / 3 / 4 in flight: 3.03 / 2.91 / 2.98 ms per commit)
        zk.MultilinearKZG.commitment_begin(poly, srs).wait()
        torch.cuda.synchronize()
        runs, total = [], 0.0
        while total < MIN_LEG_SECONDS and len(runs) < 20:
            tp = time.perf_counter()             # the pipeline fills and drains inside the timed region: `steps` whole commits
            pend, got = [], []
            for _ in range(steps):
                pend.append(zk.MultilinearKZG.commitment_begin(poly, srs))
                if len(pend) == depth:
                    got.append(pend.pop(0).wait())
            got += [h.wait() for h in pend]
            runs.append(time.perf_counter() - tp)
            total += runs[-1]
            assert len(got) == steps and all(g == com for g in got), "commitments in flight differ from the synchronous ones"
        dt_pipe = sorted(runs)[len(runs) // 2]
        pipelined = {"value": round(float(n) * steps / dt_pipe, 1), "unit": "points/s", "ms_per_commit": round(1e3 * dt_pipe / steps, 3),
                     "batches": _stats([r_ / steps for r_ in runs], 1e3, 3),
                     "in_flight": depth, "note": "zkhip_kzg_commit_begin / _end: same commitments, issued back to back"}
    # informational (SURVEY 8f rows built on the commit): SRS generation on the device and MultilinearKZG::open over the same SRS
    extras = None
    if world == 1 and log_n <= 22:
        t_s = time.perf_counter()
        zk.TrustedSetup.setup(tau)
        torch.cuda.synchronize()
        t_s = time.perf_counter() - t_s
        z = zk.Fr.synthetic(log_n, SEED_SCALARS + 0x200)
        zk.MultilinearKZG.open(poly, z, plain_srs)                 # first call derives and caches the folded SRS levels
        torch.cuda.synchronize()
        proof = zk.MultilinearKZG.open(poly, z, plain_srs)
        t_open = _timed(lambda: zk.MultilinearKZG.open(poly, z, plain_srs), torch, reps=3)
        t_o = sorted(t_open)[len(t_open) // 2]
        extras = {"srs_setup_ms": round(1e3 * t_s, 2), "open": {"ms_per_open": round(1e3 * t_o, 3), "batches": _stats(t_open, 1e3, 3), "proofs": len(proof.proofs),
                  "note": "MultilinearKZG::open (multilinear_kzg.rs:50-88): %d quotient commitments, folded SRS levels cached" % len(proof.proofs)}}
        # ... and against the level tables (shifted tables of the folded levels, built once per SRS like the commit's table)
        t_b = time.perf_counter()
        plain_srs.precompute_open()
        torch.cuda.synchronize()
        t_b = time.perf_counter() - t_b
        proof_t = zk.MultilinearKZG.open(poly, z, plain_srs)
        assert np.array_equal(proof_t.evaluation, proof.evaluation) and all(a == b for a, b in zip(proof_t.proofs, proof.proofs)), \
            "openings with and without the level tables differ"
        t_open = _timed(lambda: zk.MultilinearKZG.open(poly, z, plain_srs), torch, reps=5)
        t_o = sorted(t_open)[len(t_open) // 2]
        extras["open_level_tables"] = {"ms_per_open": round(1e3 * t_o, 3), "batches": _stats(t_open, 1e3, 3), "tables_gib": round(plain_srs.level_tables.numel() / 2 ** 30, 2),
                                       "tables_built_ms": round(1e3 * t_b, 1),
                                       "note": "the same opening against zkhip_srs_level_tables (TrustedSetup.precompute_open): same proof, checked"}
        plain_srs.invalidate()                                     # the 1.9 GiB go back before the next leg
    # commits of short polynomials (the reference's kzg bench size 2^8, kzg/benches/multilinear_kzg_benchmark.rs:10-43; plonk's callers):
    # an SRS of <= 2^12 points builds its table on first use and its commits take the short path (no sort, no buckets)
    small = None
    if world == 1:
        small = {}
        for lg in (8, 12):
            s_srs = zk.TrustedSetup.setup(zk.Fr.synthetic(lg, SEED_SCALARS + 0x400 + lg))
            s_poly = zk.Multilinear(_synthetic(zk, torch, 1 << lg, SEED_SCALARS + 0x410 + lg))
            zk.MultilinearKZG.commitment(s_poly, s_srs)
            ts_ = _timed(lambda: zk.MultilinearKZG.commitment(s_poly, s_srs), torch, reps=10)
            small["2^%d" % lg] = {"ms_per_commit": round(1e3 * sorted(ts_)[len(ts_) // 2], 4), "batches": _stats(ts_, 1e3)}
            s_z = zk.Fr.synthetic(lg, SEED_SCALARS + 0x420 + lg)
            zk.MultilinearKZG.open(s_poly, s_z, s_srs)              # (builds the SRS's folded levels and their tables on first use)
            to_ = _timed(lambda: zk.MultilinearKZG.open(s_poly, s_z, s_srs), torch, reps=10)
            small["2^%d" % lg]["ms_per_open"] = round(1e3 * sorted(to_)[len(to_) // 2], 4)
        small["note"] = ("MultilinearKZG::commitment / open on a 2^8 / 2^12-point SRS (tables built on first use): one plain sum per digit bit, two "
                         "launches -- for all rounds of an opening at once")
    # the same commitments without the table (16 instead of 13 bucket additions per point, 16 bucket reductions)
    com_plain = zk.MultilinearKZG.commitment(poly, plain_srs)
    t_plain = _timed(lambda: zk.MultilinearKZG.commitment(poly, plain_srs), torch, reps=steps)
    dt_plain = steps * sorted(t_plain)[len(t_plain) // 2]
    assert world > 1 or com_plain == com, "table and plain commitments differ"
    ctx = N.Context.get()
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "profile_enable")
    zk.MultilinearKZG.commitment(poly, srs)
    ms, cnt, by = _profile(N, ctx, b"msm_accumulate")
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "profile_enable")
    short_path = ms <= 0.0          # at most 2^12 scalars per rank (dry runs): zkhip_kzg_commit_table's short path, no accumulate pass to price
    if short_path:
        ms, by = float("nan"), 0
    out = {"metric": "MSM points/s (KZG commit, 2^%d-point SRS per GPU)" % log_n,
           "value": round(float(n) * world * steps / dt, 1), "unit": "points/s", "ms_per_commit": round(1e3 * dt / steps, 3),
           "steps": steps, "batches": _stats([b_ / steps for b_ in commit_batches], 1e3, 3),
           "sharding": ("(scalars, SRS) of one %d-point commit split over %d GPUs; one all-gather of %d partial commitments (104 B each, "
                        "device to device over RCCL) per commit, summed on every rank" % (n * world, world, world)) if world > 1 else "single GPU",
           "points_per_gpu": n, "exchanges_per_commit": 1 if world > 1 else 0, "commitment_replicated_on_all_ranks": same,
           "srs_table": {"bytes": int(srs._table.numel()), "build_ms": round(1e3 * t_tab, 1),
                         "windows": int(srs._table.numel()) // (128 * n),
                         "note": "2^(first bit of window w) * point for the digit windows of a scalar (13 of 20 / 19 bits at 2^20 points); depends on the SRS only, built once, not timed"},
           "pipelined": pipelined,
           "small_commits": small,
           "extras": extras,
           "without_srs_table": {"value": round(float(n) * steps / dt_plain, 1), "unit": "points/s (this rank)",
                                 "ms_per_commit": round(1e3 * dt_plain / steps, 3)},
           "roofline": {"bound": "integer ALU (not HBM: ~10 Fq products of ~900 instructions per bucket addition)",
                        "kernel": "msm_accumulate_kernel", "achieved": round(by / (ms * 1e-3) / 1e9, 2),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "avg_launch_us": round(1e3 * ms, 1),
                        "algorithmic_bytes_per_launch": "128 B x points (96 B affine point + 32 B scalar)"}}
    # what actually bounds that kernel: issue of v_mad_u64_u32.  One bucket addition = 10 products in the 14 x 28-bit
    # representation = 4060 multiply-adds; a launch adds one point per non-zero digit (13 windows x points with the
    # shifted-SRS table at 2^20 points, all but ~2^-19 of them).  Peak = 1024 SIMDs x 64 lanes / 4.9 cycles per wave-instruction x 2.4 GHz as measured by
    # tools/ubench.hip (profiles/r01/ubench_alu_gfx950.txt).
    windows = int(srs._table.numel()) // (128 * n)                      # 13 at 2^20 points (zkhip_srs_table_bytes)
    mads = 4060.0 * windows * n * (1.0 - 2.0 ** -19)
    peak_tmads = 1024 * 64 / 4.9 * 2.4e9 / 1e12
    if short_path:
        out["roofline"] = out["roofline_alu"] = None
        out["note_short_path"] = "commits of at most 2^12 scalars take the short path (no bucket accumulation): no accumulate-pass roofline at this size"
    else:
      out["roofline_alu"] = {"bound": "valu", "kernel": "msm_accumulate_kernel", "achieved": round(mads / (ms * 1e-3) / 1e12, 2),
                           "peak": round(peak_tmads, 2), "unit": "T v_mad_u64_u32 lane-ops/s",
                           "frac": round(mads / (ms * 1e-3) / 1e12 / peak_tmads, 4),
                           "ops_per_launch": "4060 multiply-adds x %d windows x points" % windows}
    if rank == 0 and not args.no_cpu_baseline:                         # at N > 1 too: the other ranks wait in the next leg's barrier
        from oracle import oracle as ora
        m = 1 << 14                                                     # bounded sample of the naive reference algorithm (~3.5 s)
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
                               "sample": "C oracle's naive sum of mul_bigint (multilinear_kzg.rs:43-47) on the first 2^14 "
                                         "points/scalars of the same input, %.1f s (cost is linear in points)" % cdt}
        try:   # the same naive port on every host core at once
            out["cpu_baseline"]["all_cores"] = _all_cores(_CHILD_MSM % (ROOT, 1 << 11), 1 << 11, "points/s",
                                                          "the naive commit of 2^11 points each")
        except Exception as e:
            out["cpu_baseline"]["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:   # for context only: what a CPU gets with the bucket method (not the reference's algorithm)
            mp = 1 << 17
            aff = np.zeros((mp, 13), dtype=np.uint64)
            aff[:, :12] = srs.powers_of_tau_in_g1[:mp].cpu().numpy().view(np.uint64)
            aff[:, 12] = srs.inf[:mp].cpu().numpy()
            scp = poly.evaluations[:mp].cpu().numpy().view(np.uint64)
            t1 = time.perf_counter()
            ora.msm_pippenger(scp, aff)
            pdt = time.perf_counter() - t1
            out["cpu_baseline"]["cpu_pippenger_context"] = {
                "value": round(mp / pdt, 1), "unit": "points/s", "cores": 1, "kind": "port",
                "sample": "the oracle's bucket-method MSM (NOT the reference's algorithm) on the first 2^17 points, %.1f s" % pdt}
        except Exception as e:
            out["cpu_baseline"]["cpu_pippenger_context"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def bench_composed(args, zk, N, rank, world, barrier, dist, torch, np):
    """ComposedSumcheck::prove (composed_sumcheck.rs:32-67; the reference's bench shape: a product of two tables) on
    2^composed_log_n entries per table per GPU.  N > 1: the tables are sharded by low index bits, one record of
    partial sums (96 B) all-gathered per round (zk_cryptography_amd.distributed.ShardedComposedSumcheck)."""
    from zk_cryptography_amd import distributed as D
    K, n = 2, 1 << args.composed_log_n
    tables = [_synthetic(zk, torch, n, SEED_TABLE + 16 * rank + 1 + k) for k in range(K)]
    poly = zk.ComposedMultilinear([zk.Multilinear(t) for t in tables])
    exch = [0]

    def prove():
        if world == 1:
            proof, ch = zk.ComposedSumcheck(poly).prove()
            return proof.round_polys, ch
        sh = D.ShardedComposedSumcheck(D.HipComposedEngine([tables], world, multi=False), world, comm=D.Comm.get(N.Context.get(), world, rank, dist))
        res = sh.prove()
        exch[0] = sh.exchanges
        return res

    steps = max(2, min(args.steps, 10))
    for _ in range(2):
        rp, ch = prove()
    bs, (rp, ch) = _batches(prove, steps, barrier, dist, world, torch)
    dt = sorted(bs)[len(bs) // 2]
    same = True
    if world > 1:
        same = _same_on_all_ranks(dist, torch, np, np.concatenate([np.asarray(rp, dtype=np.uint64).reshape(-1), np.asarray(ch, dtype=np.uint64).reshape(-1)]))
    return {"workload": "ComposedSumcheck::prove, product of %d tables, 2^%d entries per table per GPU" % (K, args.composed_log_n),
            "value": round(K * n * world * steps / dt, 1), "unit": "field-evals/s (table entries consumed)",
            "ms_per_prove": round(1e3 * dt / steps, 4), "batches": _stats([b_ / steps for b_ in bs], 1e3), "rounds": int(len(ch)), "steps": steps,
            "entries_per_table_per_gpu": n, "exchanges_per_prove": exch[0] if world > 1 else 0,
            "transcript_replicated_on_all_ranks": same,
            "sharding": "tables sharded by low index bits; two rounds per exchange: one record of 16 cross-block sums (+ 4 block sums) per term and rank all-gathered per stage" if world > 1 else "single GPU"}


def bench_composed_shapes(args, zk, N, torch, np):
    """The reference's OTHER composed bench shapes, one GPU: ComposedSumcheck::prove on a product of FIVE tables
    (sumcheck/benches/composed_sumcheck_benchmark.rs:33-78) and MultiComposedSumcheckProver::prove_partial on (2 + 3) tables
    (sumcheck/benches/multi_composed_sumcheck_benchmark.rs:8-54; prove_partial: the variant that does not hash the tables first).
    SURVEY 8d: 96 K N algorithmic bytes per prover."""
    out = {}
    steps = max(2, min(args.steps, 10))
    for log_n in sorted({20, args.composed_log_n}):
        n = 1 << log_n
        tabs = [zk.Multilinear(_synthetic(zk, torch, n, SEED_TABLE + 0x500 + k)) for k in range(5)]
        poly = zk.ComposedMultilinear(tabs)
        fn = lambda: zk.ComposedSumcheck(poly).prove()       # noqa: E731
        fn()
        ts = _timed(fn, torch, reps=steps)
        dt = sorted(ts)[len(ts) // 2]
        out["composed_k5_2^%d" % log_n] = {"workload": "ComposedSumcheck::prove, product of 5 tables of 2^%d entries" % log_n, "ms_per_prove": round(1e3 * dt, 4),
                                           "batches": _stats(ts, 1e3), "value": round(5 * n / dt, 1), "unit": "field-evals/s (table entries consumed)",
                                           "algorithmic_bytes": 96.0 * 5 * n, "frac_of_hbm": round(96.0 * 5 * n / dt / 1e9 / HBM_PEAK_GBS, 4),
                                           "large_rounds": "composed_round_dot_kernel (csrc/composed_dot.hpp): the last factor of every index as an int8 GEMM on the matrix cores "
                                                           "(first round from 2^18 output pairs, folding rounds from 2^19); ZKHIP_ROUND_DOT=0 = the vector form"
                                                           if os.environ.get("ZKHIP_ROUND_DOT", "1") != "0" else "vector form (ZKHIP_ROUND_DOT=0)"}
        del tabs, poly
    n = 1 << 20
    tabs = [zk.Multilinear(_synthetic(zk, torch, n, SEED_TABLE + 0x520 + k)) for k in range(5)]
    mpoly = [zk.ComposedMultilinear(tabs[:2]), zk.ComposedMultilinear(tabs[2:])]
    claimed = zk.MultiComposedSumcheckProver.calculate_poly_sum(mpoly)
    fn = lambda: zk.MultiComposedSumcheckProver.prove_partial(mpoly, claimed)       # noqa: E731
    fn()
    ts = _timed(fn, torch, reps=steps)
    dt = sorted(ts)[len(ts) // 2]
    out["multi_composed_2_3_2^20"] = {"workload": "MultiComposedSumcheckProver::prove_partial, (2 + 3) tables of 2^20 entries", "ms_per_prove": round(1e3 * dt, 4),
                                      "batches": _stats(ts, 1e3), "value": round(5 * n / dt, 1), "unit": "field-evals/s (table entries consumed)",
                                      "algorithmic_bytes": 96.0 * 5 * n, "frac_of_hbm": round(96.0 * 5 * n / dt / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def bench_gkr(args, zk, N, D, rank, world, barrier, dist, torch, np):
    """GKRProtocol::prove (gkr/src/protocol.rs:21-117) on Circuit::random(depth) -- the reference's gkr bench shape at depth 8,
    BASELINE configs[3]'s width 2^20 at depth 20.  Replicas only: every rank proves its own circuit."""
    out = {"workload": "GKRProtocol::prove on Circuit::random(depth), evaluation resident in HBM, circuit resident (zkhip_circuit)",
           "outer_transcript": "host (ZKHIP_GKR_HOST_TRANSCRIPT / ZKHIP_PIPE=0)" if (os.environ.get("ZKHIP_GKR_HOST_TRANSCRIPT", "0") not in ("", "0") or os.environ.get("ZKHIP_PIPE") == "0")
                               else "device: a hasher workgroup beside every closing kernel, no host round trip per layer",
           "replicas": world, "ms_per_proof": {}}
    for depth in (8, 20):
        circuit = zk.Circuit.random(depth)
        ev = circuit.evaluation(zk.Fr.synthetic(2 ** depth, SEED_GKR + rank))
        zk.GKRProtocol.prove(circuit, ev)
        reps = 5 if depth <= 8 else 3
        bs, _ = _batches(lambda: zk.GKRProtocol.prove(circuit, ev), reps, barrier, dist, world, torch)
        dt = sorted(bs)[len(bs) // 2] / reps
        out["ms_per_proof"]["depth_%d" % depth] = round(1e3 * dt, 3)
        out.setdefault("batches", {})["depth_%d" % depth] = _stats([b_ / reps for b_ in bs], 1e3, 3)
    if world == 1 and not args.no_gkr_threads:
        # independent proofs of one circuit from ONE C-ABI call (zkhip_gkr_prove_batch, gkr/benches/gkr_benchmark.rs:11-27 proves input after
        # input): the library runs them side by side on its internal lanes (streams, scratch and transcript state of their own; the launch
        # chains enqueued by the context's host pool) -- a proof keeps one workgroup busy most of the time, so throughput comes from independent
        # proofs.  Every proof is compared with the synchronous one.
        out["batch"] = {"note": "zkhip_gkr_prove_batch: B independent proofs of one circuit from one call from one host thread; ms per proof; every proof "
                                "equal to GKRProtocol.prove's (asserted)", "ms_per_proof": {}}
        try:
            import ctypes as C
            from zk_cryptography_amd.gkr import GKRProtocol as _G
            p_ = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
            for depth, B_list in ((8, (8, 32)), (20, (8, 24))):      # (as many proofs as lanes: the slowest lane sets the time; several per lane: handed out one by one)
                circuit = zk.Circuit.random(depth)
                evs = [circuit.evaluation(zk.Fr.synthetic(2 ** depth, SEED_GKR + 100 + b)) for b in range(max(B_list))]
                want = [[sp.to_bytes() for sp in zk.GKRProtocol.prove(circuit, ev).sumcheck_proofs] for ev in evs[:2]]
                dev = _G._device_circuit(circuit, N.Context.get())
                nl, stride = depth, 2 * depth
                for B in B_list:
                    got = zk.GKRProtocol.prove_batch(circuit, evs[:B])
                    assert [[sp.to_bytes() for sp in pr.sumcheck_proofs] for pr in got[:2]] == want, "batched and synchronous proofs differ"
                    # the C call itself, outputs preallocated once (what a Rust host pays); the Python mirror's per-proof unpacking beside it
                    ptrs = (C.c_void_p * (B * (nl + 1)))(*[t.data_ptr() for ev in evs[:B] for t in ev])
                    lens = (C.c_size_t * (nl + 1))(*[t.shape[0] for t in evs[0]])
                    o = [np.zeros((B, nl, 4), np.uint64), np.zeros((B, nl), np.uint32), np.zeros((B, nl, stride), np.uint32), np.zeros((B, nl, stride, 7, 2, 4), np.uint64),
                         np.zeros((B, nl, 4), np.uint64), np.zeros((B, nl, 4), np.uint64), np.zeros((B, 2, 4), np.uint64)]
                    call = lambda: N.check(N.lib().zkhip_gkr_prove_batch(dev.handle, C.c_uint32(B), C.c_uint32(0), ptrs, lens, *[p_(a) for a in o], None, None), "gkr_prove_batch")   # noqa: E731
                    call()
                    ts, tm = [], []
                    for _ in range(5 if depth <= 8 else 3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        call()
                        ts.append((time.perf_counter() - t0) / B)
                        t0 = time.perf_counter()
                        zk.GKRProtocol.prove_batch(circuit, evs[:B])
                        tm.append((time.perf_counter() - t0) / B)
                    out["batch"]["ms_per_proof"].setdefault("depth_%d" % depth, {})[str(B)] = round(1e3 * sorted(ts)[len(ts) // 2], 3)
                    out["batch"].setdefault("ms_per_proof_python_mirror", {}).setdefault("depth_%d" % depth, {})[str(B)] = round(1e3 * sorted(tm)[len(tm) // 2], 3)
                del evs, circuit, dev
        except Exception as e:      # noqa: BLE001 -- reported, not fatal for the other legs
            out["batch"]["error"] = repr(e)
    # BASELINE configs[3]: ONE depth-20 proof with every layer's sumcheck sharded over the ranks (GKRProtocol.prove_sharded:
    # the layer tables are built on every rank, the rounds over b and c run on shards with one record all-gathered per round)
    # next to the replicated figure above -- whichever is faster is the answer to "should GKR shard at this width"
    if world > 1 or args.force_sharded:
        depth = 20
        circuit = zk.Circuit.random(depth)
        ev = circuit.evaluation(zk.Fr.synthetic(2 ** depth, SEED_GKR))          # the same input on every rank
        gcomm = D.Comm.get(N.Context.get(), world, rank, dist if world > 1 else None)
        proof = zk.GKRProtocol.prove_sharded(circuit, ev, world, rank, comm=gcomm)
        barrier()
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            proof = zk.GKRProtocol.prove_sharded(circuit, ev, world, rank, comm=gcomm)
        barrier()
        dt = (time.perf_counter() - t0) / reps
        same = True
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            same = _same_on_all_ranks(dist, torch, np, np.concatenate([np.asarray(w, dtype=np.uint64).reshape(-1) for w in proof.wb_s + proof.wc_s]))
        want = zk.GKRProtocol.prove(circuit, ev)
        equal = all(a.to_bytes() == b.to_bytes() for a, b in zip(proof.sumcheck_proofs, want.sumcheck_proofs))
        out["sharded"] = {"workload": "one Circuit::random(20) proof, every layer's sumcheck sharded over %d GPU(s)" % world,
                          "ms_per_proof": round(1e3 * dt, 3), "exchanges_per_proof": int(proof._exchanges),
                          "proof_equals_single_gpu_proof": bool(equal), "proof_replicated_on_all_ranks": same,
                          "compare_with": "ms_per_proof.depth_20 (every rank proving the whole circuit by itself)"}
    return out


def bench_h2d(args, zk, N, table, poly, torch, np):
    """PCIe-inclusive figures: the reference's Multilinear owns a host Vec<F> (evaluation_form.rs:6-9), so a literal drop-in that keeps
    nothing resident uploads the table (512 MiB at 2^24) before every proof and the scalars (32 MiB at 2^20) before every commit.  Never
    `value`: tables and SRS stay resident in the product."""
    out = {}
    n = len(poly)
    host = torch.empty_like(table, device="cpu").pin_memory()
    host.copy_(table)
    dev = torch.empty_like(table)
    p2 = zk.Multilinear(dev)

    def step():
        dev.copy_(host, non_blocking=True)
        sc = zk.Sumcheck(p2)
        sc.poly_sum()
        return sc.prove()

    ts = _timed(step, torch, reps=3, min_total=0.2, max_batches=8)
    med = sorted(ts)[len(ts) // 2]
    out["sumcheck"] = {"ms_per_step": round(1e3 * med, 3), "value": round(n / med, 1), "unit": "field-evals/s",
                       "upload_bytes": int(n * 32), "upload_GBps": None, "batches": _stats(ts, 1e3, 3),
                       "note": "pinned host table -> HBM (hipMemcpyAsync) + poly_sum + prove per step"}
    t_up = _timed(lambda: dev.copy_(host, non_blocking=True), torch, reps=3, min_total=0.1, max_batches=6)
    out["sumcheck"]["upload_GBps"] = round(n * 32 / sorted(t_up)[len(t_up) // 2] / 1e9, 1)
    del host, dev, p2
    return out


def bench_strong_and_config4(args, zk, N, rank, world, barrier, dist, torch, np):
    """N > 1 only.  (a) STRONG scaling of the headline: ONE 2^log_n-entry table over all ranks (2^log_n / N entries per GPU), where the
    default line is weak scaling (2^log_n per GPU).  (b) BASELINE configs[4]'s shape: a multilinear KZG commit of 2^26 evaluations
    sharded 8-way = 2^23 points per GPU (here: 2^23 per GPU at any N), one all-gather of the partial commitments."""
    from zk_cryptography_amd import distributed as D
    comm = D.Comm.get(N.Context.get(), world, rank, dist)
    out = {}
    n_total = 1 << args.log_n
    if n_total // world >= 1 << 12:
        shard = _synthetic(zk, torch, n_total // world, SEED_TABLE + 0x400 + rank)
        ex = [0]

        def step():
            sh = D.ShardedSumcheck(D.HipSumcheckEngine(shard), world, comm=comm)
            r = sh.prove()
            ex[0] = sh.exchanges
            return r

        for _ in range(3):
            step()
        bs, res = _batches(step, args.steps, barrier, dist, world, torch)
        med = sorted(bs)[len(bs) // 2]
        out["sumcheck_strong"] = {"workload": "ONE 2^%d-entry table over %d GPUs (2^%d entries per GPU)" % (args.log_n, world, (n_total // world).bit_length() - 1),
                                  "value": round(n_total * args.steps / med, 1), "unit": "field-evals/s", "ms_per_step": round(1e3 * med / args.steps, 4),
                                  "batches": _stats([b / args.steps for b in bs], 1e3), "exchanges_per_prove": ex[0], "scaling": "strong"}
        del shard
    # (b) ONE SRS for the whole commit: the same tau on every rank, generated whole on every GPU (6 GiB at 2^26, 0.6 s) and cut down to
    # this rank's points g, g + N, ...; this rank's scalars are entries g, g + N, ... of ONE polynomial p.  Checked once before the
    # timing: commit == p(tau) * G, with p(tau) = sum_g eq_g(tau_low) * p_g(tau_high) from every rank's evaluation of its shard.
    log_c = args.config4_log_n
    log_w = world.bit_length() - 1
    tau = zk.Fr.synthetic(log_c + log_w, SEED_SCALARS + 0x300)                     # variables 0 .. log_c - 1 index j (high bits), the rest the rank
    whole = zk.TrustedSetup.setup(tau)
    srs = zk.TrustedSetup(whole.powers_of_tau_in_g1[rank::world].contiguous(), whole.inf[rank::world].contiguous())
    del whole
    torch.cuda.empty_cache()
    srs.precompute()
    poly = zk.Multilinear(_synthetic(zk, torch, 1 << log_c, SEED_SCALARS + 0x310 + rank))

    def commit():
        return D.sharded_commit(None, srs.inf, poly.evaluations, comm, table=srs.table)

    com = commit()
    # the identity
    R_MOD, to_int = zk.Fr.MODULUS, (lambda a: zk.Fr.to_ints(np.asarray(a, dtype=np.uint64).reshape(1, 4))[0])
    mine = torch.from_numpy(np.ascontiguousarray(poly.evaluation(tau[:log_c])).view(np.int64).copy()).cuda()
    every = torch.empty(4 * world, dtype=torch.int64, device="cuda")
    dist.all_gather_into_tensor(every, mine)
    vals = every.cpu().numpy().view(np.uint64).reshape(world, 4)
    t_low = [to_int(t) for t in tau[log_c:]]
    p_tau = 0
    for g in range(world):
        w = 1
        for i, t in enumerate(t_low):                                               # variable log_c + i <-> bit (log_w - 1 - i) of g
            w = w * (t if (g >> (log_w - 1 - i)) & 1 else (1 - t)) % R_MOD
        p_tau = (p_tau + w * to_int(vals[g])) % R_MOD
    one_var = zk.TrustedSetup.setup(zk.Fr.synthetic(1, SEED_SCALARS + 0x3FF))      # [G (1 - x), G x]: committing [v, v] against it is v * G
    want = zk.MultilinearKZG.commitment(zk.Multilinear(zk.Fr.from_ints([p_tau, p_tau])), one_var)
    identity = bool((not com[1]) and (not want.infinity) and np.array_equal(np.asarray(com[0], dtype=np.uint64), want.xy))
    if not identity:
        sys.stderr.write("bench.py: rank %d: sharded commit != p(tau) * G on ONE sharded SRS\n" % rank)
        sys.exit(3)
    bs, com = _batches(commit, 3, barrier, dist, world, torch, min_total=0.2, max_batches=6)
    med = sorted(bs)[len(bs) // 2]
    same = _same_on_all_ranks(dist, torch, np, np.concatenate([np.asarray(com[0], dtype=np.uint64), np.array([1 if com[1] else 0], dtype=np.uint64)]))
    out["commit_config4_shape"] = {"workload": "multilinear KZG commit of 2^%d evaluations, (scalars, SRS) of ONE commit sharded over %d GPUs: 2^%d points per GPU" % (log_c + log_w, world, log_c),
                                   "value": round(float(1 << log_c) * world * 3 / med, 1), "unit": "points/s", "ms_per_commit": round(1e3 * med / 3, 3),
                                   "batches": _stats([b / 3 for b in bs], 1e3, 3), "exchanges_per_commit": 1, "commitment_replicated_on_all_ranks": same,
                                   "commit_equals_p_tau_times_G": identity,
                                   "identity_note": "one SRS (same tau on every rank, rank g holds points g, g + N, ...), p(tau) from every rank's evaluation of its shard, p(tau) * G as a two-entry commit; checked before the timing, exit 3 on mismatch"}
    return out


def bench_prediction(args, zk, N, torch, np, dist_mod, exchange):
    """N = 1 only: what the sharded paths cost WITHOUT their exchanges -- the protocols run with world = 1 on inputs of the size one rank
    of an 8-GPU job holds -- so that t(N) = t_local + exchanges x exchange_us is a prediction the first real multi-GPU run can be checked
    against.  exchange_us for 8 ranks cannot be measured here: the measured one-rank cost of the call path plus an
    ASSUMED 15 us for an 8-rank RCCL all-gather of <= 64 KiB over xGMI."""
    from zk_cryptography_amd import distributed as D
    x1 = exchange["64_B"]["back_to_back_us"] if exchange and "64_B" in exchange else None
    x64k = exchange["65536_B"]["back_to_back_us"] if exchange and "65536_B" in exchange else None
    xh = exchange["64_B"]["with_host_wait_us"] if exchange and "64_B" in exchange else None
    assumed_fabric_us = 15.0
    out = {"model": "t(N) = t_local(shard) + exchanges(N) x exchange_us; t_local measured here with world = 1 on one rank's share, exchange_us = "
                    "measured one-rank call path + %.0f us ASSUMED for an 8-rank RCCL all-gather of <= 64 KiB over xGMI" % assumed_fabric_us,
           "exchange_us_world_1": {"64_B": x1, "65536_B": x64k, "64_B_with_host_wait": xh}, "assumed_fabric_us_8_ranks": assumed_fabric_us, "n_gpus": 8}
    if x1 is None:
        return out
    x8, x8k, x8h = x1 + assumed_fabric_us, x64k + assumed_fabric_us, xh + assumed_fabric_us

    def local_sumcheck(log_shard):
        t = _synthetic(zk, torch, 1 << log_shard, SEED_TABLE + 0x500)
        ex = [0]

        def step():
            sh = D.ShardedSumcheck(D.HipSumcheckEngine(t), 1)
            sh.prove()
            ex[0] = sh.exchanges
        ts = _timed(step, torch, reps=5, min_total=0.15, max_batches=20)
        return sorted(ts)[len(ts) // 2], ex[0]

    t24, e24 = local_sumcheck(args.log_n)
    t21, e21 = local_sumcheck(args.log_n - 3)
    out["sumcheck_weak_2^%d_per_gpu" % args.log_n] = {"t_local_ms": round(1e3 * t24, 4), "exchanges": 3,
                                                      "predicted_ms_per_step": round(1e3 * t24 + 2e-3 * x8 + 1e-3 * x8k, 4),
                                                      "predicted_field_evals_per_s": round(8.0 * (1 << args.log_n) / (t24 + 2e-6 * x8 + 1e-6 * x8k), 1)}
    out["sumcheck_strong_2^%d_total" % args.log_n] = {"t_local_ms": round(1e3 * t21, 4), "exchanges": 3,
                                                     "predicted_ms_per_step": round(1e3 * t21 + 2e-3 * x8 + 1e-3 * x8k, 4),
                                                     "predicted_field_evals_per_s": round(float(1 << args.log_n) / (t21 + 2e-6 * x8 + 1e-6 * x8k), 1)}
    # composed prover: K = 2, 2^composed_log_n per table per GPU; at 8 ranks three more exchanged rounds than the one-rank run
    K, nloc = 2, 1 << args.composed_log_n
    tabs = [_synthetic(zk, torch, nloc, SEED_TABLE + 0x510 + k) for k in range(K)]
    ex = [0]

    def cstep():
        sh = D.ShardedComposedSumcheck(D.HipComposedEngine([tabs], 1, multi=False), 1, use_stages=True)
        sh.prove()
        ex[0] = sh.exchanges
    ts = _timed(cstep, torch, reps=3, min_total=0.15, max_batches=20)
    tc = sorted(ts)[len(ts) // 2]
    e8 = ex[0] + 2              # three more rounds are exchanged at 8 ranks (the table is 8 times larger): two exchanges with two rounds per exchange
    out["composed_2x2^%d_per_gpu" % args.composed_log_n] = {"t_local_ms": round(1e3 * tc, 4), "exchanges": e8, "predicted_ms_per_prove": round(1e3 * tc + 1e-3 * e8 * x8, 4)}
    del tabs
    # GKR, Circuit::random(20): one proof sharded over 8 ranks against one proof per rank (replicas)
    circuit = zk.Circuit.random(20)
    ev = circuit.evaluation(zk.Fr.synthetic(1 << 20, SEED_GKR))
    tr = _timed(lambda: zk.GKRProtocol.prove(circuit, ev), torch, reps=2, min_total=0.15, max_batches=8)
    exg = [0]

    def gstep():
        exg[0] = zk.GKRProtocol.prove_sharded(circuit, ev, 1, 0, use_stages=True)._exchanges     # two rounds per exchange, as the ranks of a real job run it
    tg = _timed(gstep, torch, reps=1, min_total=0.15, max_batches=8)
    t_rep, t_sh = sorted(tr)[len(tr) // 2], sorted(tg)[len(tg) // 2]
    # exchanges: a session exchanges one record per round while (local entries x world) exceeds the tail and then gathers once -- a count
    # that depends on the LAYER's width, not on the rank count; the one-rank run counts them (layers narrower than 16 values, which 8
    # ranks prove unsharded, included: an upper bound by a handful)
    e8g = exg[0]
    out["gkr_depth_20"] = {"replicas_ms_per_proof": round(1e3 * t_rep, 3), "sharded_t_local_ms": round(1e3 * t_sh, 3), "sharded_exchanges_8_ranks": e8g,
                           "sharded_predicted_ms_per_proof": round(1e3 * t_sh + 1e-3 * e8g * x8, 3),
                           "expected_winner": "replicas" if t_rep < t_sh + 1e-6 * e8g * x8 else "sharded",
                           "note": "the sharded t_local is the WHOLE layer on one rank (upper bound of an 8-rank rank's share: table building and "
                                   "streaming rounds shrink 8-fold, the serial rounds and launches do not)"}
    # commit, BASELINE configs[4]: 2^23 points per GPU + one 104-byte all-gather whose result the host reads
    out["commit_2^23_per_gpu"] = {"exchanges": 1, "exchange_us_with_host_wait": round(x8h, 1),
                                  "note": "t_local = the 2^23-point commit of `msm` run with --msm-log-n 23 (21.9 ms when last measured); predicted = t_local + exchange"}
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
    ap.add_argument("--ntt-log-n", type=int, default=21, help="log2 of the transform size of the NTT leg")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-fold", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="diagnostic: run the sharded prover protocol even on one GPU")
    ap.add_argument("--no-h2d", action="store_true")
    ap.add_argument("--no-gkr-threads", action="store_true", help="skip the GKR batch leg (zkhip_gkr_prove_batch: B proofs of one circuit side by side)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the proofs-in-flight leg (profiling: keeps the kernel averages those of the synchronous steps)")
    ap.add_argument("--config4-log-n", type=int, default=23, help="N > 1: log2 of the per-GPU points of the configs[4]-shaped commit (2^26 over 8 = 2^23)")
    ap.add_argument("--no-exchange", action="store_true", help="skip the exchange-cost measurement and the N = 8 prediction")
    ap.add_argument("--selftest", action="store_true", help="run the sharded provers' self-test (always on at N > 1) on one GPU too")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        # plain `python bench.py --gpus N`: become the launcher (nothing has touched the GPU yet) -- one process per GPU
        import subprocess
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d -- launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d "
                 "bench.py --gpus %d ...)" % (args.gpus, world, args.gpus, args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

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
    table = _synthetic(zk, torch, n, SEED_TABLE + 16 * rank)      # uniform field elements
    poly = zk.Multilinear(table)

    from zk_cryptography_amd import distributed as D
    exchanges = [0]
    # the communicator of the sharded provers: the library's own RCCL communicator on the `nccl` group (the unique id travels through the
    # group once), a host-staged callback in the one-GPU dry run, none on one rank
    comm = D.Comm.get(N.Context.get(), world, rank, dist if world > 1 else None)
    exchange = None
    if world > 1 or args.selftest:
        # ---- self-test before anything is timed: the measured exchange first, then one small proof of every sharded prover against
        # rank 0 and against the single-GPU prover; a mismatch ends the job (non-zero status on every rank)
        if world > 1 and not args.no_exchange:
            exchange = bench_exchange(comm, world)
        st = selftest(zk, np, torch, N, D, comm, rank, world, dist, (lambda a: _same_on_all_ranks(dist, torch, np, a)) if world > 1 else (lambda a: True))
        if rank == 0:
            print("bench.py selftest: " + json.dumps({"exchange": exchange, "n_gpus": world, "sharded_provers_match_single_gpu_and_rank_0": st}), file=sys.stderr, flush=True)
        if not all(st.values()):
            sys.stderr.write("bench.py: rank %d: sharded self-test FAILED: %r\n" % (rank, st))
            sys.exit(3)

    def step():
        if world == 1 and not args.force_sharded:
            sc = zk.Sumcheck(poly)
            sc.poly_sum()
            return sc.prove()
        # N > 1: ONE prover over the world * 2^log_n-entry table whose rank-interleaved shard is `table`
        # (overlapped stage: two all-gathers of block sums -- the second one beside the shard's fold -- and one of the 256-entry
        # local tables, over RCCL/xGMI, replicated transcript)
        sh = D.ShardedSumcheck(D.HipSumcheckEngine(table), world, comm=comm)
        res = sh.prove()
        exchanges[0] = sh.exchanges
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = step()
    # batches of exactly `steps` steps, each between barrier + synchronize, until >= 0.3 s are measured; the line reports the MEDIAN batch
    # (a 20-step batch is 7 ms: box-to-box and run-to-run spread is larger than most single changes) and min / max beside it
    batch_s, res = _batches(step, args.steps, barrier, dist, world, torch)
    dt = sorted(batch_s)[len(batch_s) // 2]
    step_stats = _stats([b / args.steps for b in batch_s], 1e3)
    # ---- roofline of the dominant kernel: HIP events around every fold launch on the launch stream
    # (BEFORE the proofs-in-flight leg: once its lanes exist -- four more low-priority fold streams beside the synchronous prover's on four hardware queues --
    # the events around the big fold span ~7 us more than the kernel runs (91.5 against 84.9 us; rocprofv3's kernel trace: 80.4 us either way).  The
    # kernels are measured in the state the timed steps above ran in.)
    ctx = N.Context.get()
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 1), "profile_enable")
    prof_steps = max(1, min(args.steps, 5))
    for _ in range(prof_steps):
        step()
    ms, cnt, by = _profile(N, ctx, b"multifold")
    per_kernel = {}
    step_bytes = 0.0
    for kname, label in ((b"fine_sums", "fine_sums_kernel (poly_sum: sums of every run of 256 entries, 32 n B)"),
                         (b"multifold", "multifold_mfma_kernel<4, 4> (k-variable fold, 32 (n + n / 2^k) B)"),
                         (b"chunk_sums", "chunk_sums_kernel (block sums of the generic plan, 32 n B)"),
                         (b"multifold_small", "multifold_kernel<16> (k-variable fold, few outputs)"), (b"blockfold", "blockfold_kernel (L2-resident)")):
        kms, kcnt, kby = _profile(N, ctx, kname)
        step_bytes += kby / prof_steps
        if kcnt and kby > 0 and kname in (b"fine_sums", b"multifold", b"chunk_sums"):
            kgbs = kby / (kms * 1e-3) / 1e9
            per_kernel[kname.decode()] = {"kernel": label, "achieved": round(kgbs, 1), "frac": round(kgbs / HBM_PEAK_GBS, 4), "launches": kcnt,
                                          "avg_launch_us": round(1e3 * kms / kcnt, 2), "bytes_per_launch": kby / kcnt}
    N.check(N.lib().zkhip_profile_enable(ctx.handle, 0), "profile_enable")
    # informational: the same steps with several proofs in flight (zkhip_sumcheck_prove_begin / _end) -- the way a prover with
    # several tables calls the library; the device runs the proofs in stream order, the idle time between them shrinks
    pipelined = None
    if world == 1 and not args.force_sharded and not args.no_pipelined:
        # several tables, proved round robin with up to `depth` proofs in flight: every ticket has its own streams and buffers
        # (zkhip_ctx::ProofLane), so the streaming passes of one table's proof run while the transcript rounds of the others hash
        n_tab = 8
        polys = [poly] + [zk.Multilinear(_synthetic(zk, torch, n, SEED_TABLE + 16 * rank + 8 + t)) for t in range(1, n_tab)]
        want = [res]
        for pl in polys[1:]:
            sc0 = zk.Sumcheck(pl); sc0.poly_sum(); want.append(sc0.prove())

        def in_flight(k, depth):
            pend, got = [], [None] * n_tab
            for i in range(k):
                sc = zk.Sumcheck(polys[i % n_tab])
                sc.poly_sum()
                pend.append((i % n_tab, sc.prove_begin()))
                if len(pend) == depth:
                    j, h = pend.pop(0)
                    got[j] = h.wait()
            for j, h in pend:
                got[j] = h.wait()
            torch.cuda.synchronize()
            return got

        by_depth = {}
        n_run = max(args.steps, 96)             # proofs per timed run: the pipeline fills and drains inside it (~4 proofs' worth at eight in flight)
        for depth in (2, 3, 4, 6, 8):
            in_flight(2 * depth, depth)         # the lanes' streams and buffers come into being on first use
            runs, total = [], 0.0
            while total < MIN_LEG_SECONDS / 2 and len(runs) < 40:
                t1 = time.perf_counter()
                got = in_flight(n_run, depth)
                runs.append(time.perf_counter() - t1)
                total += runs[-1]
            by_depth[depth] = (sorted(runs)[len(runs) // 2], runs)
            for g_, w_ in zip(got, want):
                assert g_ is None or (np.array_equal(g_[1], w_[1]) and np.array_equal(g_[0].univariate_poly, w_[0].univariate_poly)), "in-flight and synchronous proofs differ"
        depth = min(by_depth, key=lambda d: by_depth[d][0])
        dtp = by_depth[depth][0]
        pipelined = {"value": round(float(n) * n_run / dtp, 1), "unit": "field-evals/s", "ms_per_step": round(1e3 * dtp / n_run, 4), "in_flight": depth,
                     "tables": n_tab, "proofs_per_run": n_run, "batches": _stats([r / n_run for r in by_depth[depth][1]], 1e3),
                     "ms_per_step_by_in_flight": {str(d): round(1e3 * by_depth[d][0] / n_run, 4) for d in sorted(by_depth)},
                     "note": "zkhip_sumcheck_prove_begin / _end, %d tables round robin, %d proofs per timed run (fill and drain inside); every ticket has workspace "
                             "and scratch of its own, from the third proof in flight on every streaming pass runs on the caller's stream (the big folds "
                             "held back behind the next tables' sums passes); proofs bit-identical to the synchronous ones (asserted)" % (n_tab, n_run)}
        del polys, sc0, by_depth, got
    transcript_same = True
    if world > 1:
        s_, rp_, ch_ = res
        transcript_same = _same_on_all_ranks(dist, torch, np, np.concatenate([np.asarray(s_, dtype=np.uint64).reshape(-1),
                                                                                  np.asarray(rp_, dtype=np.uint64).reshape(-1),
                                                                                  np.asarray(ch_, dtype=np.uint64).reshape(-1)]))
        assert transcript_same, "rank %d: the sharded proof differs from rank 0's" % rank

    achieved = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters: separate rocprofv3 --pmc passes of this command, committed under profiles/rNN
    # (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); the NEWEST round's file is read and named -- a file constant of that
    # round's build, labelled as such, not a counter of this run
    traffic, traffic_source, traffic_all, traffic_age = None, None, None, None
    try:
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "pmc_traffic.json")))
        if files and args.log_n == 24 and world == 1:
            pmc = json.load(open(files[-1]))
            src = os.path.relpath(files[-1], ROOT)
            traffic = pmc["multifold"]["hbm_bytes_per_launch"]
            traffic_all = {k: v.get("hbm_bytes_per_launch") for k, v in pmc.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}
            for short, needle in (("fine_sums", "fine_sums_kernel"), ("chunk_sums", "chunk_sums_kernel")):
                hits = [v for k, v in pmc.get("kernels", {}).items() if needle in k and "hbm_bytes_per_launch" in v]
                if hits and short not in traffic_all:
                    traffic_all[short] = max(h["hbm_bytes_per_launch"] for h in hits)      # the 2^24-entry launch of the step
            traffic_source = src + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in round %s; not collected in this run)" % src.split(os.sep)[1]
            # how old is that constant?  The file carries a fingerprint of the streaming kernels' sources it was collected with (tools/pmc_summary.py);
            # compared with the tree this run comes from -- a rebuilt kernel must not silently keep the old ratio (tools/collect.sh refreshes it)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from pmc_summary import kernel_sources_fingerprint
            now_fp, then_fp = kernel_sources_fingerprint(ROOT), pmc.get("kernel_sources_sha16")
            traffic_age = {"collected_with_kernel_sources_sha16": then_fp, "this_run_kernel_sources_sha16": now_fp,
                           "stale": (then_fp != now_fp) if then_fp else None,
                           "note": ("the PMC file predates the fingerprint: age unknown" if not then_fp else
                                    "same kernel sources as the PMC passes" if then_fp == now_fp else
                                    "the streaming kernels' sources CHANGED since the PMC passes: re-collect (tools/collect.sh)")}
    except Exception:
        traffic = None
    longest = max(per_kernel, key=lambda k: per_kernel[k]["avg_launch_us"]) if per_kernel else None
    # the line's roofline entry = the LONGEST streaming kernel of the step (the dominant one); the others are under per_kernel
    lk = per_kernel.get(longest) if longest else None
    if lk is not None:
        achieved, cnt = lk["achieved"], lk["launches"]
        ms = lk["avg_launch_us"] * cnt / 1e3
        if traffic_all and longest in traffic_all:
            traffic = traffic_all[longest]
    roofline = {"bound": "hbm", "kernel": lk["kernel"] if lk else "multifold_mfma_kernel<4, 4>",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "traffic_age": traffic_age,
                "launches": cnt, "avg_launch_us": round(1e3 * ms / max(1, cnt), 2),
                "algorithmic_bytes_per_launch": lk["bytes_per_launch"] if lk else None,
                "algorithmic_bytes_note": "fine_sums: 32 B x table entries read; multifold: 32 B x (table entries read + folded entries written), k variables per launch",
                # every streaming kernel of a step against the same roof, and the step as a whole on the bytes it REALLY moves
                "per_kernel": per_kernel, "longest_streaming_kernel": longest, "traffic_per_kernel": traffic_all,
                "step_bytes_moved": step_bytes,
                "step_hbm_frac": round(step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if dt > 0 else None,
                "step_hbm_frac_note": "bytes the step's kernels really move (their algorithmic bytes, summed) / ms_per_step / 8 TB/s: the proof is "
                                      "latency-bound by its Fiat-Shamir rounds between and beside the two streaming passes"}

    def leg(skip, fn, *a):
        if skip:
            return None
        if world > 1:
            return fn(*a)        # several ranks: a leg with collectives that failed on ONE rank would leave the others blocked -- let torchrun end the job
        try:
            return fn(*a)
        except Exception as e:   # reported, not hidden: the headline legs above are already measured
            return {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- the single-variable fold (48 n bytes), rank-local
    fold = leg(args.no_fold, bench_fold, args, zk, N, poly, torch)
    # ---- full evaluation (96 n bytes algorithmic), rank-local
    evaluate = leg(args.no_fold, bench_evaluate, args, zk, N, poly, torch)
    # ---- second half of BASELINE's metric: MSM points/s of the KZG commit on a 2^20-point SRS per GPU
    msm = None
    if not args.no_msm:
        msm = bench_msm(args, zk, N, rank, world, barrier, dist, torch, np)
    # ---- NTT / polynomial product (rank-local: replicas)
    ntt = leg(args.no_ntt, bench_ntt, args, zk, N, torch)
    # ---- the composed prover (GKR's sumcheck shape) on sharded tables; informational, never part of `value`
    composed = leg(args.no_composed, bench_composed, args, zk, N, rank, world, barrier, dist, torch, np)
    composed_shapes = leg(args.no_composed or world > 1, bench_composed_shapes, args, zk, N, torch, np)
    gkr = leg(args.no_gkr, bench_gkr, args, zk, N, D, rank, world, barrier, dist, torch, np)
    # ---- PCIe-inclusive figure (never `value`), exchange cost and the N = 8 prediction (N = 1), strong scaling + configs[4] shape (N > 1)
    h2d = leg(args.no_h2d or world > 1, bench_h2d, args, zk, N, table, poly, torch, np)
    prediction = multi = None
    if not args.no_exchange and world == 1:
        # one rank: the call path's fixed cost over a ONE-RANK RCCL communicator of the library's own (librccl resolved at run time)
        try:
            # (RCCL prints a version banner on STDOUT when its first communicator comes up: stdout carries ONE JSON line, so the banner goes to stderr)
            sys.stdout.flush()
            _fd1 = os.dup(1)
            os.dup2(2, 1)
            try:
                rc1 = D.Comm(N.Context.get(), 1, 0, transport="rccl")
            finally:
                try:
                    import ctypes as _C
                    _C.CDLL(None).fflush(None)          # the banner sits in the C library's stdio buffer: out with it while fd 1 is stderr
                except Exception:                        # noqa: BLE001
                    pass
                os.dup2(_fd1, 1)
                os.close(_fd1)
            exchange = bench_exchange(rc1, 1)
            rc1.close()
        except Exception as e:
            exchange = {"error": "%s: %s" % (type(e).__name__, e)}
        prediction = leg(False, bench_prediction, args, zk, N, torch, np, None, exchange if exchange and "error" not in exchange else None)
    if world > 1:
        multi = bench_strong_and_config4(args, zk, N, rank, world, barrier, dist, torch, np)

    # ---- CPU baseline: the oracle's single-threaded restatement of poly_sum + prove, rank 0 only.  Last: the all-cores leg
    # loads every host core, which would disturb the host-side share of the GPU legs above if it ran before them
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:      # at N > 1 too: the other ranks wait in the barrier below
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
            cpu["all_cores"] = _all_cores(_CHILD % (ROOT, 2), 2 * (1 << 22), "field-evals/s", "2 runs of the same C port on a 2^22-entry table each")
        except Exception as e:
            cpu["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if world > 1:
        barrier()                        # the ranks that ran no CPU baseline wait here for rank 0
    if rank == 0:
        total_evals = float(n) * world * args.steps
        if world > 1:                    # which transport the sharded provers ran on, and why if it is not the library's RCCL communicator
            import ctypes as C
            ver = C.c_int(0)
            ok = N.lib().zkhip_rccl_version(C.byref(ver)) == 0
            exchange = dict(exchange or {})
            exchange.update({"transport": comm.transport if isinstance(comm.transport, str) else "callback", "fallback_reason": comm.fallback_reason,
                             "rccl_version": ver.value if ok else None, "torch_nccl_version": list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None})

        def g(d, *path):
            for k in path:
                if not isinstance(d, dict) or k not in d:
                    return None
                d = d[k]
            return d
        # every leg's headline figure, flat and FIRST in the line (a record that keeps only the head or the tail of the line keeps these)
        legs = {"step_ms": round(1e3 * dt / args.steps, 4), "step_hbm_frac": roofline.get("step_hbm_frac"),
                "pipelined_ms": g(pipelined, "ms_per_step"),
                "fine_sums_us": g(per_kernel, "fine_sums", "avg_launch_us"), "fine_sums_frac": g(per_kernel, "fine_sums", "frac"),
                "kfold_us": g(per_kernel, "multifold", "avg_launch_us"), "kfold_frac": g(per_kernel, "multifold", "frac"),
                "fold_ms": g(fold, "ms_per_fold"), "fold_frac": g(fold, "roofline", "frac"),
                "evaluate_ms": g(evaluate, "ms_per_evaluation"), "evaluate_frac_96n": g(evaluate, "algorithmic_frac_of_hbm"), "evaluate_pass_frac": g(evaluate, "roofline", "frac"),
                "msm_ms": g(msm, "ms_per_commit"), "msm_mpoints_s": round(g(msm, "value") / 1e6, 1) if g(msm, "value") else None,
                "msm_alu_frac": g(msm, "roofline_alu", "frac"), "msm_inflight_ms": g(msm, "pipelined", "ms_per_commit"),
                "msm_no_table_ms": g(msm, "without_srs_table", "ms_per_commit"),
                "commit_2^8_ms": g(msm, "small_commits", "2^8", "ms_per_commit"), "commit_2^12_ms": g(msm, "small_commits", "2^12", "ms_per_commit"),
                "open_ms": g(msm, "extras", "open", "ms_per_open"), "open_tables_ms": g(msm, "extras", "open_level_tables", "ms_per_open"),
                "open_2^8_ms": g(msm, "small_commits", "2^8", "ms_per_open"), "open_2^12_ms": g(msm, "small_commits", "2^12", "ms_per_open"),
                "srs_setup_ms": g(msm, "extras", "srs_setup_ms"),
                "ntt_ms": g(ntt, "ms_per_fft"), "intt_ms": g(ntt, "ms_per_ifft"), "multiply_ms": g(ntt, "ms_per_multiply"), "ntt_alu_frac": g(ntt, "roofline_alu", "frac"),
                "composed_k2_ms": g(composed, "ms_per_prove"),
                "composed_k5_2^20_ms": g(composed_shapes, "composed_k5_2^20", "ms_per_prove"), "composed_k5_2^20_frac": g(composed_shapes, "composed_k5_2^20", "frac_of_hbm"),
                "composed_k5_2^%d_ms" % args.composed_log_n: g(composed_shapes, "composed_k5_2^%d" % args.composed_log_n, "ms_per_prove"),
                "composed_k5_2^%d_frac" % args.composed_log_n: g(composed_shapes, "composed_k5_2^%d" % args.composed_log_n, "frac_of_hbm"),
                "multi_composed_2_3_ms": g(composed_shapes, "multi_composed_2_3_2^20", "ms_per_prove"), "multi_composed_2_3_frac": g(composed_shapes, "multi_composed_2_3_2^20", "frac_of_hbm"),
                "gkr8_ms": g(gkr, "ms_per_proof", "depth_8"), "gkr20_ms": g(gkr, "ms_per_proof", "depth_20"),
                "gkr8_batch8_ms": g(gkr, "batch", "ms_per_proof", "depth_8", "8"), "gkr8_batch32_ms": g(gkr, "batch", "ms_per_proof", "depth_8", "32"),
                "gkr20_batch8_ms": g(gkr, "batch", "ms_per_proof", "depth_20", "8"), "gkr20_batch24_ms": g(gkr, "batch", "ms_per_proof", "depth_20", "24"), "gkr20_sharded_ms": g(gkr, "sharded", "ms_per_proof"),
                "h2d_step_ms": g(h2d, "sumcheck", "ms_per_step"),
                "cpu_1core_mevals_s": round(g(cpu, "value") / 1e6, 2) if g(cpu, "value") else None,
                "cpu_msm_1core_points_s": g(msm, "cpu_baseline", "value"),
                "strong_ms": g(multi, "sumcheck_strong", "ms_per_step"), "config4_commit_ms": g(multi, "commit_config4_shape", "ms_per_commit"),
                "transport": g(exchange, "transport") if world > 1 else None}
        if composed_shapes is not None and composed is not None:
            composed = dict(composed)
            composed["other_shapes"] = composed_shapes
        out = {
            "legs": legs,
            "metric": "field-evals/s (sumcheck 2^%d) + MSM points/s (KZG 2^%d) per BASELINE.json; value = the sumcheck prover's "
                      "field-evals/s, the MSM half is under \"msm\"" % (args.log_n, args.msm_log_n),
            "value": round(total_evals / dt, 1),
            "unit": "field-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "batches": {**step_stats, "unit": "ms per step", "note": "batches of exactly `steps` steps between barrier + synchronize; value / ms_per_step = the median batch"},
            "pipelined": pipelined,
            "higher_is_better": True,
            "scaling": "weak",
            **({"dry_run": "ZKHIP_BENCH_ONE_GPU: all ranks on one GPU, exchange over gloo -- not a measurement"} if one_gpu else {}),
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"field": "BLS12-381 Fr (255-bit Montgomery, 8 x u32 limbs); G1 over Fq (381-bit)",
                       "workload": "24-var multilinear sumcheck prover (poly_sum + prove), BLS12-381 Fr" if args.log_n == 24
                       else "%d-var multilinear sumcheck prover" % args.log_n,
                       "evals_per_gpu": n,
                       "inputs": "uniform field elements, splitmix64-seeded xoshiro256** (SURVEY 8d), table seed 0x5EED000000000001 + 16 rank",
                       "exchanges_per_prove": exchanges[0] if world > 1 else 0,
                       "transcript_replicated_on_all_ranks": transcript_same,
                       "sharding": ("one %d-entry table sharded by low index bits over %d GPUs; overlapped stage: RCCL all-gathers of the coarse block sums, of the folded fine sums (beside the shard's local fold) and of the 256-entry local tables; replicated transcript" % (n * world, world))
                       if world > 1 else "single GPU"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "fold": fold,
            "evaluate": evaluate,
            "msm": msm,
            "ntt": ntt,
            "composed": composed,
            "gkr": gkr,
            "h2d_inclusive": h2d,
            "exchange": exchange,
            **({"prediction_n8": prediction} if prediction is not None else {}),
            **({"multi_gpu": multi} if multi is not None else {}),
        }
        print(json.dumps(out))
    D.Comm.close_all()                   # RCCL communicators before the process group that carried their id
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

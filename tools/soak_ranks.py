"""Soak of what four processes sharing ONE GPU exposed in round 5 (a wrong GKR proof on one rank: the outer transcript's hasher read a
round's items from behind another L2 before they were there; fixed by an agent-scope release in outer_publish, csrc/composed_kernels.hpp):
`world` processes, each proving Circuit::random(depth) `repeats` times with the device-resident outer transcript AND through the sharded
prover over gloo, every proof compared bit for bit with the first one / with rank 0's.
usage (GPU box): python tools/soak_ranks.py [world=4] [repeats=50] [depths=9,16]      exit code 0 = every proof identical"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def worker(rank, world, port, repeats, depths, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import zk_cryptography_amd as zk
        torch.cuda.set_device(0)
        bad, n = [], 0
        for depth in depths:
            circuit = zk.Circuit.random(depth)
            ev = circuit.evaluation(zk.Fr.random(2 ** depth, 300 + depth))
            want = None
            for rep in range(repeats):
                proof = zk.GKRProtocol.prove(circuit, ev)                              # the outer transcript on the device
                sharded = zk.GKRProtocol.prove_sharded(circuit, ev, world, rank, None, dist)
                sig = [sp.to_bytes() for sp in proof.sumcheck_proofs] + [np.asarray(w).tobytes() for w in proof.wb_s + proof.wc_s]
                sig_s = [sp.to_bytes() for sp in sharded.sumcheck_proofs] + [np.asarray(w).tobytes() for w in sharded.wb_s + sharded.wc_s]
                if want is None:
                    want = sig
                if sig != want or sig_s != want:
                    bad.append((depth, rep, sig != want, sig_s != want))
                n += 2
        # every rank must hold the same proofs
        digest = np.frombuffer(__import__("hashlib").sha256(b"".join(want)).digest(), dtype=np.uint8).copy()
        t = torch.from_numpy(digest)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        same = all(bool((g == gathered[0]).all()) for g in gathered)
        q.put((rank, bad, n, same))
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    depths = [int(d) for d in sys.argv[3].split(",")] if len(sys.argv) > 3 else [9, 16]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=worker, args=(r, world, port, repeats, depths, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1500) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    ok = all(not bad and same for _, bad, _, same in res) and all(p.exitcode == 0 for p in procs)
    print("soak: %d processes on one GPU x %d repeats x depths %s: %d proofs per process, %s" %
          (world, repeats, depths, res[0][2], "ALL IDENTICAL" if ok else "MISMATCH %r" % [(r, b, s) for r, b, _, s in res]))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

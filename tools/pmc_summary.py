"""Post-processing of the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each with --kernel-trace only) of bench.py:
per-kernel HBM bytes per launch, FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM section: the counter
reports half of a wide coalesced streaming read), WRITE_SIZE as is (exact for 16-byte streaming stores).  The counters are KB.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources of the streaming kernels whose traffic is reported: bench.py compares this fingerprint with the tree it runs from
# (`roofline.traffic_age`), so that a rebuilt kernel does not silently keep an old ratio
STREAMING_KERNEL_SOURCES = ("mle_kernels.hpp", "multifold_kernels.hpp", "mfma_fold.hpp", "fp.hpp", "fp_mul_gen.hpp", "wide_acc.hpp")


def kernel_sources_fingerprint(root=ROOT):
    h = hashlib.sha256()
    for f in STREAMING_KERNEL_SOURCES:
        h.update(open(os.path.join(root, "zk-cryptography_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc["%s grid=%s" % (name, r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python3 bench.py --steps 3 --warmup 1 "
                     "--no-cpu-baseline --no-msm --no-composed --no-gkr --no-ntt --no-h2d --no-exchange` on MI355X; FETCH_SIZE doubled per the gfx950 correction; units of the "
                     "counters: KB; hbm_bytes_per_launch = 2 * FETCH + WRITE",
           "kernel_sources_sha16": kernel_sources_fingerprint(), "kernel_sources": list(STREAMING_KERNEL_SOURCES), "kernels": {}}
    for k in sorted(fetch):
        if not k.startswith("zk::"):
            continue
        out["kernels"][k] = {"launches": nf[k], "FETCH_SIZE_KB_avg": fetch[k], "WRITE_SIZE_KB_avg": write.get(k, 0.0),
                             "hbm_bytes_per_launch": int(round((2 * fetch[k] + write.get(k, 0.0)) * 1024))}
    # the prover's k-variable fold (writes its outputs) and `evaluation`'s pass (the WSUM form: third template argument true)
    big = [k for k in out["kernels"] if ("multifold_mfma_kernel" in k or "multifold_kernel" in k) and ", true>" not in k]
    if big:
        k = max(big, key=lambda q: out["kernels"][q]["hbm_bytes_per_launch"])
        out["multifold"] = dict(out["kernels"][k], kernel=k)
    ev = [k for k in out["kernels"] if "multifold_mfma_kernel" in k and ", true>" in k]
    if ev:
        k = max(ev, key=lambda q: out["kernels"][q]["hbm_bytes_per_launch"])
        out["multifold_eval"] = dict(out["kernels"][k], kernel=k)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()

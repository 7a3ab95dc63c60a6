"""Reads bench.py's JSON line on stdin and prints the figures one looks at first."""
import json, sys
for l in sys.stdin:
    l = l.strip()
    if not l.startswith("{"):
        if l and "amdgpu.ids" not in l: print(l[:400])
        continue
    d = json.loads(l)
    g = lambda o, *ks: (g(o.get(ks[0]), *ks[1:]) if len(ks) > 1 else o.get(ks[0])) if isinstance(o, dict) else None
    print("n_gpus", d["n_gpus"], "ms_per_step", d["ms_per_step"], "pipelined", g(d, "pipelined", "ms_per_step"), "value %.3e" % d["value"], "| multifold", g(d, "roofline", "avg_launch_us"), "us frac", g(d, "roofline", "frac"),
          "traffic", g(d, "roofline", "traffic"))
    for k in ("fold", "msm", "ntt", "composed", "gkr", "cpu_baseline"):
        v = d.get(k)
        if v is None: continue
        if "error" in v: print(" ", k, "ERROR", v["error"]); continue
        if k == "fold": print("  fold ms", v["ms_per_fold"], "frac", v["roofline"]["frac"], "us", v["roofline"]["avg_launch_us"])
        if k == "msm": print("  msm ms", v["ms_per_commit"], "pipelined", g(v, "pipelined", "ms_per_commit"), "plain", g(v, "without_srs_table", "ms_per_commit"), "alu frac", g(v, "roofline_alu", "frac"),
                             "cpu", g(v, "cpu_baseline", "value"), g(v, "cpu_baseline", "all_cores", "value"), g(v, "cpu_baseline", "cpu_pippenger_context", "value"), "| srs", g(v, "extras", "srs_setup_ms"), "open", g(v, "extras", "open", "ms_per_open"))
        if k == "ntt": print("  ntt fft", v["ms_per_fft"], "ifft", v["ms_per_ifft"], "multiply", v["ms_per_multiply"], "hbm frac", v["roofline"]["frac"], "valu frac", v["roofline_alu"]["frac"])
        if k == "composed": print("  composed ms", v["ms_per_prove"], "exch", v.get("exchanges_per_prove"), "same", v.get("transcript_replicated_on_all_ranks"))
        if k == "gkr": print("  gkr", v["ms_per_proof"])
        if k == "cpu_baseline": print("  cpu", v["value"], g(v, "all_cores", "value"))

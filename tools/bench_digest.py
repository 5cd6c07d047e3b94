#!/usr/bin/env python3
"""One short line per bench.py JSON line on stdin (value, step, kernels, roofline fraction, placement report)."""
import json
import sys

for ln in sys.stdin:
    ln = ln.strip()
    if not ln.startswith("{"):
        if ln.startswith("k_bin_hist ms per step"):
            v = [float(x) for x in ln.split(":")[1].split()]
            print("  K1 per step: first %.3f min %.3f max %.3f last %.3f" % (v[0], min(v), max(v), v[-1]))
        continue
    d = json.loads(ln)
    km = d["kernels_ms"]
    print("value %.1f %s  step %.4f ms  K1 %.4f ms  frac %.3f  %s  n_gpus %d  launch: %s  allreduce probe %s  placement %s" % (
        d["value"], d["unit"], d["ms_per_step"], km["k_bin_hist"], d["roofline"]["frac"],
        "  ".join("%s %.4f" % (k, v) for k, v in km.items() if k != "k_bin_hist"), d["n_gpus"],
        d["config"].get("step_launch"), d.get("allreduce_probe"), (d.get("placement") or {}).get("experiment")))
    for name, c in (d.get("configs") or {}).items():
        print("  config %-6s %s" % (name, {k: c[k] for k in ("job_ms", "value", "phases_ms", "ms_per_Mbins", "error") if k in c}))

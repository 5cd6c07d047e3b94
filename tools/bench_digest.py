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
    print("value %.1f %s  step %.3f ms  K1 %.3f ms  frac %.3f  rest %.3f  n_gpus %d  placement %s" % (
        d["value"], d["unit"], d["ms_per_step"], d["kernels_ms"]["k_bin_hist"], d["roofline"]["frac"],
        list(d["kernels_ms"].values())[1], d["n_gpus"], d.get("placement")))

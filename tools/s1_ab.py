#!/usr/bin/env python3
"""GPU box: bench.py's S1 headline (no extras) under a list of settings, one child process each, twice through the list:
usage: s1_ab.py "lib=<file under tools/_ab_libs>,KEY=VALUE,..." ...   ("-" = the in-tree library, no setting)."""
import json, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for spec in sys.argv[1:] * 2:
    env = dict(os.environ)
    if spec != "-":
        for kv in spec.split(","):
            k, v = kv.split("=", 1)
            if k == "lib":
                env["EPILOGOS_HIP_LIB"] = R + "/tools/_ab_libs/" + v
            else:
                env[k] = v
    r = subprocess.run([sys.executable, R + "/bench.py", "--no-cpu-baseline", "--configs", "none", "--dist-variants", "0", "--graph-leg", "0", "--placement-experiment", "0"],
                       env=env, capture_output=True, text=True)
    try:
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        sh = d["s1_paths"]["shard_1875000_bins"]
        print("%-30s %.1f Mbins/s  step %.4f ms  K1 %.4f  rest %.4f | shard step %.4f K1 %.4f | placement %s" % (
            spec, d["value"], d["ms_per_step"], d["kernels_ms"]["k_bin_hist"], d["kernels_ms"]["combine(normalise,table)+score_from_hist"],
            sh["session_ms_per_step"], sh["session_k_bin_hist_ms"], (d["placement"].get("report") or {}).get("whole_matrix_ms")), flush=True)
    except Exception as e:
        print(spec, "failed", r.stderr[-400:])

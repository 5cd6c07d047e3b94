#!/usr/bin/env python3
"""GPU box: bench.py's s3_small job (2 M bins x 833 x 18) under a list of settings, one child process each:
usage: s3_ab.py "lib=<file under tools/_ab_libs>,KEY=VALUE,..." ...   ("-" = the in-tree library, no setting)."""
import json, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for spec in sys.argv[1:]:
    env = dict(os.environ)
    if spec != "-":
        for kv in spec.split(","):
            k, v = kv.split("=", 1)
            if k == "lib":
                env["EPILOGOS_HIP_LIB"] = R + "/tools/_ab_libs/" + v
            else:
                env[k] = v
    r = subprocess.run([sys.executable, R + "/bench.py", "--no-cpu-baseline", "--configs", "s3_small", "--dist-variants", "0", "--graph-leg", "0", "--placement-experiment", "0",
                        "--shard-bins", "0", "--steps", "2", "--warmup", "1", "--config-reps", "3", "--s3-small-bins", os.environ.get("S3_AB_BINS", "2000000")], env=env, capture_output=True, text=True)
    try:
        p = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["configs"]["s3_small"]
        print("%-40s job %.2f ms  phases %s" % (spec, p["job_ms"], p["phases_ms"]), flush=True)
    except Exception as e:
        print(spec, "failed", r.stderr[-400:])

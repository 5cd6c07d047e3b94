#!/usr/bin/env python3
"""GPU box: bench.py's paired job (BASELINE config 5) under a list of environment settings, one child process each:
usage: paired_sweep.py "K=V,K=V" "K=V" ...   ("-" = no setting).  Prints job_ms and the three phase times."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for spec in sys.argv[1:] or ["-"]:
    env = dict(os.environ)
    if spec != "-":
        for kv in spec.split(","):
            k, v = kv.split("=", 1)
            env[k] = v
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--configs", "paired", "--dist-variants", "0", "--graph-leg", "0",
                        "--placement-experiment", "0", "--shard-bins", "0", "--steps", "3", "--warmup", "1", "--config-reps", "5"], env=env, capture_output=True, text=True)
    try:
        p = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["configs"]["paired"]
        print("%-60s job %.3f ms  phases %s  host %.2f" % (spec, p["job_ms"], [round(v, 3) for v in p["phases_ms"].values()], p["host_enqueue_ms_of_the_count_phase"]), flush=True)
    except Exception as e:
        print(spec, "failed", repr(e), r.stderr[-500:], flush=True)

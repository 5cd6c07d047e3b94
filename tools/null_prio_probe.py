#!/usr/bin/env python3
"""How do the null sampler (second stream) and the count passes share the chip?  The paired job of bench.py for a list of
EPG_NULL_BLOCKS_PER_CU values (the sampler's persistent grid; default 8), and -- `--priority` -- for the stream priorities HIP
offers (EPILOGOS_NULL_PRIORITY 1 / 0 / -1).  Round 4: neither moves it (5.2-5.3 ms; fewer sampler blocks per CU are slower)."""
import ctypes as C, json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(0), C.c_int(0)
print("hipDeviceGetStreamPriorityRange ->", hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "least", lo.value, "greatest", hi.value)
var, values = ("EPILOGOS_NULL_PRIORITY", (None, "1", "0", "-1")) if "--priority" in sys.argv else ("EPG_NULL_BLOCKS_PER_CU", (None, "1", "2", "3", "4", "6"))
for prio in values:
    env = dict(os.environ)
    env.pop(var, None)
    if prio is not None:
        env[var] = prio
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--configs", "paired", "--no-cpu-baseline", "--placement-experiment", "0", "--shard-bins", "0",
                        "--steps", "5", "--warmup", "2", "--config-reps", "5"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(prio, "failed", r.stderr[-300:]); continue
    v = json.loads(line[-1])["configs"]["paired"]
    print(var, prio, "job_ms", v.get("job_ms"), list(v.get("phases_ms", {}).values()), v.get("error"))

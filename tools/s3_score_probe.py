#!/usr/bin/env python3
"""Times the S3 score call alone (kernel + table build + transpose) at N = 833, S = 18 for a list of EPG_S3_SCORE_DBG values --
each in a child process, because the library reads the variable once.  usage: s3_score_probe.py [--bins 1048576] [--dbg 0,1,2,4,6]"""
import argparse
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
ap = argparse.ArgumentParser()
ap.add_argument("--bins", type=int, default=1048576)
ap.add_argument("--dbg", default="0,1,2,8,4,6")
ap.add_argument("--child", action="store_true")
a = ap.parse_args()
if not a.child:
    for d in a.dbg.split(","):
        env = dict(os.environ, EPG_S3_SCORE_DBG=d)
        r = subprocess.run([sys.executable, __file__, "--child", "--bins", str(a.bins)], env=env, capture_output=True, text=True)
        print("EPG_S3_SCORE_DBG=%s: %s" % (d, (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1]), flush=True)
    sys.exit(0)
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402
N, S, R = 833, 18, a.bins
X = torch.empty((R, N), dtype=torch.int8, device="cuda")
bench.generate_shard(torch, X, N, S, 0)
q = torch.rand((N, N, S, S), device="cuda", dtype=torch.float32)
q /= q.sum()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
engine.score_s3(X, N, S, q)
torch.cuda.synchronize()
ev[0].record()
for _ in range(2):
    engine.score_s3(X, N, S, q)
ev[1].record()
torch.cuda.synchronize()
print("%.2f ms per call" % (ev[0].elapsed_time(ev[1]) / 2))

#!/usr/bin/env python3
"""A/B of the S3 contraction's LDS ring depth (EPG_S3_RING=3|4, read once per process): one child process per setting, the
expected pass on --bins bins x 833 biosamples, median of the timed calls and a checksum of the counts.
usage: s3_ring_ab.py [--bins 1048576] [--reps 3]"""
import argparse
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def child(bins, reps):
    import numpy as np
    import torch
    sys.path.insert(0, str(ROOT))
    import bench
    from epilogos_amd import engine
    engine.require_gpu()
    S, N = 18, 833
    X = engine.alloc_states(bins, N)
    bench.generate_shard(torch, X, N, S, 0)
    c3 = torch.zeros(N * N * S * S, dtype=torch.int32, device="cuda")
    ws = None
    ts = []
    for i in range(reps + 1):
        c3.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.hist_s3(X, N, S, counts=c3); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    w = (torch.arange(c3.numel(), device="cuda", dtype=torch.int64) % 1000003) + 1
    print("ring %s syrk %s dbg %s: %.3f ms (all: %s) checksum %d total %d" % (os.environ.get("EPG_S3_RING", "default"), os.environ.get("EPG_S3_SYRK", "") or "default", os.environ.get("EPG_S3_DBG", "0"), float(np.median(ts[1:])),
          " ".join("%.2f" % t for t in ts), int((c3.long() * w).sum().item()), int(c3.long().sum().item())), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--variants", default="3::,4::,4:pp:,3::,4::,4:pp:", help="comma list of ring:syrk:dbg (EPG_S3_RING, EPG_S3_SYRK, EPG_S3_DBG)")
    a = ap.parse_args()
    if a.child:
        child(a.bins, a.reps)
    else:
        for v in a.variants.split(","):
            ring, syrk, dbg = (v.split(":") + ["", ""])[:3]
            env = dict(os.environ, EPG_S3_RING=ring or "3", EPG_S3_SYRK=syrk, EPG_S3_DBG=dbg or "0")
            subprocess.run([sys.executable, __file__, "--child", "--bins", str(a.bins), "--reps", str(a.reps)], env=env, check=False)

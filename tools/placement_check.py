#!/usr/bin/env python3
"""GPU box: engine.alloc_hist's policy in a FRESH process -- fourteen S1 jobs through the session on one resident matrix (job 1 plain,
job 2 the quick search, job 4 the deep one if the plain allocation was kept), the wall time of every job, the search's report, then
k_bin_hist over the whole matrix with the histogram cache where the jobs now have it and at a fresh plain allocation.
usage: placement_check.py [--bins 15000000] [--procs 5]   (--procs > 1: that many child processes, one after the other;
profiles/r05a_placement_spread.txt)"""
import argparse
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def one(bins, N, S):
    import numpy as np
    import torch
    from epilogos_amd import backend, engine
    import bench
    dev = torch.device("cuda", 0)
    X = engine.alloc_states(bins, N, device=dev)
    bench.generate_shard(torch, X, N, S, 0)
    torch.cuda.synchronize()
    counts = torch.zeros(S, dtype=torch.int64, device=dev)

    def k1(H, reps=5):
        ts = []
        for k in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            engine.bin_hist(X, N, S, counts=counts, H=H, want_hist=H is not None)
            e1.record()
            torch.cuda.synchronize()
            if k:
                ts.append(e0.elapsed_time(e1))
        return round(float(np.median(ts)), 4)

    import time
    be = backend.HipBackend(device=dev)
    ts, per_job = [], []
    for k in range(14):                                          # the product's policy: job 1 plain, job 2 the quick search, job 4 the deep one if needed
        sess = be.open_single(S, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pid = sess.add_device(X, N)
        sess.ensure_acc(N)
        sess.finish_device(bins, N)
        o = sess.scores_device(pid)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        per_job.append(round(ts[-1], 2))
        del o, sess
    rep = engine.placement_report()
    H = engine.alloc_hist(X, N, S)                               # where the jobs' cache now lies (the home, or a plain allocation)
    out = {"job_ms": per_job, "session_job_ms_median_of_last_8": round(float(np.median(ts[6:])), 4), "report": rep,
           "k1_counts_only_ms": k1(None), "k1_home_ms": k1(H)}
    del H
    out["k1_plain_ms"] = k1(torch.empty((bins, S), dtype=torch.int16, device=dev))
    out["frac_home"] = round(bins * N / out["k1_home_ms"] / 1e6 / 8000, 4)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", type=int, default=15_000_000)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child or a.procs == 1:
        one(a.bins, 833, 18)
    else:
        for k in range(a.procs):
            subprocess.call([sys.executable, __file__, "--bins", str(a.bins), "--child"])

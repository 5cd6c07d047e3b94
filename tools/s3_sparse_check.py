#!/usr/bin/env python3
"""GPU box: the modal-state S3 score kernel (epg_s3_sparse.hip, EPG_S3_SCORE=s) against the dense biosample-lane kernel
(EPG_S3_SCORE=l) -- same fixed-point unit, so the float64 scores must be IDENTICAL -- on a list of shapes, then both timed
at N = 833 on chr1-like synthetic states.  usage: s3_sparse_check.py [--bins 1000000] [--skip-small]"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bins", type=int, default=1_000_000)
ap.add_argument("--skip-small", action="store_true")
ap.add_argument("--skip-big", action="store_true")
a = ap.parse_args()
engine.require_gpu()


def run(kind, X, N, S, q, want64=True):
    os.environ["EPG_S3_SCORE"] = kind
    try:
        return engine.score_s3(X, N, S, q, want32=True, want64=want64)
    finally:
        os.environ.pop("EPG_S3_SCORE", None)


if not a.skip_small:
    rng = np.random.default_rng(3)
    shapes = [(18, 33, 200), (18, 64, 1920), (18, 65, 1921), (18, 129, 4000), (15, 40, 300), (20, 97, 2500), (2, 5, 77), (7, 31, 129),
              (18, 200, 3841), (11, 32, 128)]
    for S, N, R in shapes:
        p = rng.dirichlet(np.full(S, 0.3))
        p[int(rng.integers(0, S))] += 2.0
        p /= p.sum()
        x = rng.choice(S, size=(R, N), p=p).astype(np.int8)
        if R > 3 and N > 2:
            x[rng.integers(0, R), rng.integers(0, N)] = -1
            x[rng.integers(0, R), rng.integers(0, N)] = 31
        q = rng.random((N, N, S, S)).astype(np.float32) ** 3
        q[rng.random(q.shape) < 0.05] = 0.0
        q /= q.sum()
        X = engine.states_to_device(x)
        qd = torch.from_numpy(q.reshape(-1)).cuda()
        d32, d64 = run("l", X, N, S, qd)
        s32, s64 = run("s", X, N, S, qd)
        torch.cuda.synchronize()
        same = torch.equal(d64, s64) and torch.equal(d32, s32)
        err = float((d64 - s64).abs().max())
        print("S=%2d N=%3d R=%5d  identical=%s  max|diff|=%.3g  max|score|=%.3g" % (S, N, R, same, err, float(d64.abs().max())), flush=True)
        if not same:
            bad = torch.nonzero((d64 != s64).any(dim=1)).flatten()
            print("   differing bins: %d, first %s" % (bad.numel(), bad[:8].tolist()), flush=True)

if not a.skip_big:
    S, N, R = 18, 833, a.bins
    X = engine.alloc_states(R, N)
    bench.generate_shard(torch, X, N, S, 0)
    c3 = engine.hist_s3(X, N, S)
    q3 = engine.normalise(c3)
    del c3
    ws = engine.workspace(3, R, N, S)
    res = {}
    for kind in ("l", "s", "s", "l"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        os.environ["EPG_S3_SCORE"] = kind
        o32, _ = engine.score_s3(X, N, S, q3, want32=True, want64=False, ws=ws)
        torch.cuda.synchronize()
        os.environ.pop("EPG_S3_SCORE")
        print("kind %s: %.2f ms per call (%d bins)" % (kind, (time.perf_counter() - t0) * 1e3, R), flush=True)
        res.setdefault(kind, o32.clone())
    print("N=833 identical float32 scores:", torch.equal(res["l"], res["s"]), " max|diff| %.3g" % float((res["l"] - res["s"]).abs().max()))

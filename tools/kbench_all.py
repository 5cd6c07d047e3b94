#!/usr/bin/env python3
"""Timing of the non-headline kernels at BASELINE.json shapes (documentation numbers for DESIGN.md; not the contract
bench).  usage: kbench_all.py [--bins 15000000] [--s3-bins 4096] [--null-bins 1000000]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bins", type=int, default=15_000_000)
ap.add_argument("--s3-bins", type=int, default=4096)
ap.add_argument("--null-bins", type=int, default=1_000_000)
ap.add_argument("--what", default="s2,s3,null")
a = ap.parse_args()
engine.require_gpu()
S, N = 18, 833


def timed(fn, reps=3):
    ts = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:])), out


if "s2" in a.what:
    R = a.bins
    X = engine.alloc_states(R, N)
    bench.generate_shard(torch, X, N, S, 0)
    H, counts = engine.bin_hist(X, N, S)
    t, c2 = timed(lambda: engine.hist_s2_from_binhist(H, S))
    print("S2 expected from H : %8.3f ms  %.2f Gbins/s" % (t, R / t / 1e6))
    q2 = engine.normalise(c2 // 4)   # c2 accumulated over the 4 timed calls; the ratio is what matters
    o32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
    ws = engine.workspace(2, 0, N, S)
    t, _ = timed(lambda: engine.score_s2_from_binhist(H, N, S, q2, out32=o32, ws=ws))
    print("S2 score from H    : %8.3f ms  %.2f Gbins/s  (%.1f G pair-terms/s)" % (t, R / t / 1e6, R * 324 / t / 1e6))
    del X, H, o32
    torch.cuda.empty_cache()
if "s3" in a.what:
    R = a.s3_bins
    X = engine.alloc_states(R, N)
    bench.generate_shard(torch, X, N, S, 0)
    c3 = torch.zeros(N * N * S * S, dtype=torch.int32, device="cuda")
    t, _ = timed(lambda: engine.hist_s3(X, N, S, counts=c3), reps=2)
    pairs = R * N * (N - 1)
    print("S3 expected        : %8.3f ms for %d bins  -> %.1f bins/s, %.3g pair-increments/s" % (t, R, R / t * 1e3, pairs / t * 1e3))
    q3 = engine.normalise(c3)
    t, _ = timed(lambda: engine.score_s3(X, N, S, q3), reps=2)
    print("S3 score           : %8.3f ms for %d bins  -> %.1f bins/s, %.3g pair-terms/s (incl. 899 MB table build)" % (t, R, R / t * 1e3, pairs / t * 1e3))
    del X, c3, q3
    torch.cuda.empty_cache()
if "null" in a.what:
    R, NA, NB = a.null_bins, 379, 342
    XA, XB = engine.alloc_states(R, NA), engine.alloc_states(R, NB)
    bench.generate_shard(torch, XA, NA, S, 0)
    bench.generate_shard(torch, XB, NB, S, 0)
    t, _ = timed(lambda: engine.null_hist(XA, NA, XB, NB, S, NA, NB, seed=1))
    print("paired null shuffle: %8.3f ms for %d bins x (%d+%d) -> %.2f Mbins/s" % (t, R, NA, NB, R / t / 1e3))
if "pair" in a.what:
    R, NA, NB = a.bins, 379, 342
    sa = torch.randn((R, S), dtype=torch.float32, device="cuda")
    sb = torch.randn((R, S), dtype=torch.float32, device="cuda")
    t, (delta, _) = timed(lambda: engine.pair_finish(sa, sb))
    print("pair_finish        : %8.3f ms for %d bins -> %.2f Gbins/s (%.0f GB/s of its 3*72+4 B/bin)" % (t, R, R / t / 1e6, R * 220 / t / 1e6))
    t, _ = timed(lambda: engine.pair_metrics(delta, roundtrip=True))
    print("pair_metrics       : %8.3f ms for %d bins -> %.2f Gbins/s (%.0f GB/s of its 72+8 B/bin)" % (t, R, R / t / 1e6, R * 80 / t / 1e6))
    del sa, sb, delta
    torch.cuda.empty_cache()
    XA, XB = engine.alloc_states(R, NA), engine.alloc_states(R, NB)
    bench.generate_shard(torch, XA, NA, S, 0)
    bench.generate_shard(torch, XB, NB, S, 0)
    t, _ = timed(lambda: engine.quiescent(XA, NA, XB, NB, S - 1))
    print("quiescent          : %8.3f ms for %d bins x (%d+%d) -> %.2f Gbins/s (%.0f GB/s)" % (t, R, NA, NB, R / t / 1e6, R * (NA + NB) / t / 1e6))
if "null" in a.what:
    R, NA, NB = a.bins, 379, 342
    XA, XB = engine.alloc_states(R, NA), engine.alloc_states(R, NB)
    bench.generate_shard(torch, XA, NA, S, 0)
    bench.generate_shard(torch, XB, NB, S, 0)
    HA, _ = engine.bin_hist(XA, NA, S, want_counts=False)
    HB, _ = engine.bin_hist(XB, NB, S, want_counts=False)
    t, _ = timed(lambda: engine.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=1))
    print("null groups from H : %8.3f ms for %d bins x (%d+%d) -> %.2f Gbins/s" % (t, R, NA, NB, R / t / 1e6))
    t, _ = timed(lambda: engine.null_hist(XA, NA, XB, NB, S, NA, NB, seed=1), reps=1)
    print("null groups from X : %8.3f ms for %d bins x (%d+%d) -> %.2f Gbins/s" % (t, R, NA, NB, R / t / 1e6))

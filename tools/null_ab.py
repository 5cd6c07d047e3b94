#!/usr/bin/env python3
"""A/B of the histogram-based null sampler (EPG_NULL_HIST unset = bit-string kernel, seq = the round-2 kernel): 15 M bins x
(379 + 342) columns at the chr1 state frequencies, default group sizes and -g 100.  usage: null_ab.py [--bins 15000000]"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bins", type=int, default=15_000_000)
a = ap.parse_args()
engine.require_gpu()
S, NA, NB, R = 18, 379, 342, a.bins
XA, XB = engine.alloc_states(R, NA), engine.alloc_states(R, NB)
bench.generate_shard(torch, XA, NA, S, 0)
bench.generate_shard(torch, XB, NB, S, 0)
HA, _ = engine.bin_hist(XA, NA, S, want_counts=False)
HB, _ = engine.bin_hist(XB, NB, S, want_counts=False)
del XA, XB


def timed(fn, reps=3):
    ts = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:])), out


for ga, gb in ((NA, NB), (100, 100)):
    for mode in ("", "seq", "", "seq"):
        if mode:
            os.environ["EPG_NULL_HIST"] = mode
        else:
            os.environ.pop("EPG_NULL_HIST", None)
        t, (OA, OB) = timed(lambda: engine.null_hist_from_binhist(HA, HB, NA + NB, S, ga, gb, seed=1))
        sa, sb = OA.long().sum(dim=1), OB.long().sum(dim=1)
        ok = bool((sa == ga).all()) and bool((sb == gb).all()) and bool(((OA.long() + OB.long()) <= (HA.long() + HB.long())).all())
        mean17 = float(OA[:, 17].double().mean())
        print("groups %3d + %3d  %-10s %8.3f ms  sums ok %s  mean A[17] %.4f (expected %.4f)" % (
            ga, gb, mode or "bit-string", t, ok, mean17, float((HA[:, 17].double() + HB[:, 17].double()).mean()) * ga / (NA + NB)), flush=True)

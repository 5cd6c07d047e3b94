#!/usr/bin/env python3
"""Do the HBM-bound count pass (k_bin_hist) and the VALU-bound null sampler (k_null_hist_h) overlap when they run on two streams?
Times each alone and both together (the sampler on histograms of other bins than the count pass reads), for the paired shapes
(379 + 342 columns) and for the count pass on the 833-column matrix.  usage: overlap_probe.py [--bins 7500000]"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bins", type=int, default=7_500_000)
a = ap.parse_args()
engine.require_gpu()
S, NA, NB, R = 18, 379, 342, a.bins
XA, XB = engine.alloc_states(R, NA), engine.alloc_states(R, NB)
bench.generate_shard(torch, XA, NA, S, 0)
bench.generate_shard(torch, XB, NB, S, 0)
HA, _ = engine.bin_hist(XA, NA, S, want_counts=False)
HB, _ = engine.bin_hist(XB, NB, S, want_counts=False)
HA2, HB2 = torch.empty_like(HA), torch.empty_like(HB)           # what the count passes write while the sampler reads HA, HB
cA, cB = engine.zeros_counts(S, device=XA.device), engine.zeros_counts(S, device=XA.device)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def counts():
    engine.bin_hist(XA, NA, S, counts=cA, H=HA2)
    engine.bin_hist(XB, NB, S, counts=cB, H=HB2)


def sampler():
    engine.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=1)


def timed(fn_a, fn_b, reps=5):
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s1.wait_event(e0); s2.wait_event(e0)
        if fn_a:
            with torch.cuda.stream(s1):
                fn_a()
        if fn_b:
            with torch.cuda.stream(s2):
                fn_b()
        torch.cuda.current_stream().wait_stream(s1)
        torch.cuda.current_stream().wait_stream(s2)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


ta, tb, tab = timed(counts, None), timed(None, sampler), timed(counts, sampler)
print("%d bins x (%d + %d): two count passes %.3f ms, null sampler %.3f ms, both on two streams %.3f ms (sum %.3f, max %.3f)" % (
    R, NA, NB, ta, tb, tab, ta + tb, max(ta, tb)), flush=True)
tba = timed(sampler, counts)
print("   sampler enqueued first: %.3f ms" % tba, flush=True)


def counts_only():
    engine.bin_hist(XA, NA, S, counts=cA, want_hist=False)
    engine.bin_hist(XB, NB, S, counts=cB, want_hist=False)


def each(fn_a, fn_b, reps=5):
    """Duration of fn_a on s1 and of fn_b on s2 when both start together (events per stream)."""
    out = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s1.wait_event(e0); s2.wait_event(e0)
        with torch.cuda.stream(s1):
            fn_a(); ea.record()
        with torch.cuda.stream(s2):
            fn_b(); eb.record()
        torch.cuda.synchronize()
        out.append((e0.elapsed_time(ea), e0.elapsed_time(eb)))
    return tuple(float(np.median([o[k] for o in out[1:]])) for k in (0, 1))


tc = timed(counts_only, None)
a1, b1 = each(counts_only, sampler)
a2, b2 = each(counts, sampler)
print("   counts-only passes (a pure read stream) alone %.3f ms; with the sampler beside them: reads %.3f ms, sampler %.3f ms" % (tc, a1, b1), flush=True)
print("   count passes with the H store beside the sampler: count %.3f ms, sampler %.3f ms" % (a2, b2), flush=True)

#!/usr/bin/env python3
"""Within-process interleaved A/B timing of the streaming kernels (tuning aid, not the contract bench).
usage: kbench.py [--bins 15000000] [--rounds 7] variant:blocks_per_cu ...   e.g. 0:8 1:8 1:4"""
import argparse
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from epilogos_amd import _abi, engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bins", type=int, default=15_000_000)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--biosamples", type=int, default=833)
ap.add_argument("--packed", action="store_true")
ap.add_argument("--what", default="hist,score")
ap.add_argument("configs", nargs="*", default=["0:8", "1:8"])
a = ap.parse_args()

engine.require_gpu()
lib = _abi.load()
lib.epg_test_force.argtypes = [ctypes.c_int32, ctypes.c_int32]
N, S, R = a.biosamples, 18, a.bins
if a.packed:
    flat = torch.empty(R * N + 64, dtype=torch.int8, device="cuda")
    X = flat[:R * N].view(R, N)
else:
    X = engine.alloc_states(R, N)
bench.generate_shard(torch, X, N, S, 0)
H = torch.empty((R, S), dtype=torch.int16, device="cuda")
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
out32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
q = torch.full((S,), 1.0 / S, dtype=torch.float32, device="cuda")
ws = engine.workspace(1, 0, N, S)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


res = {}
for rnd in range(a.rounds + 1):
    for cfg in a.configs:
        v, bpc = [int(t) for t in cfg.split(":")]
        lib.epg_test_force(4, bpc)
        if "hist" in a.what:
            counts.zero_()
            t = timed(lambda: engine.bin_hist(X, N, S, counts=counts, H=H))
            if rnd:
                res.setdefault((cfg, "bin_hist"), []).append(t)
        if "nowrite" in a.what:
            counts.zero_()
            t = timed(lambda: engine.bin_hist(X, N, S, counts=counts, want_hist=False))
            if rnd:
                res.setdefault((cfg, "hist_noH"), []).append(t)
        if "fromhist" in a.what:
            t = timed(lambda: engine.score_s1_from_binhist(H, N, S, q, out32=out32, ws=ws))
            if rnd:
                res.setdefault((cfg, "score_fromH"), []).append(t)
        if "score" in a.what.split(","):
            t = timed(lambda: engine.score_s1(X, N, S, q, out32=out32, ws=ws))
            if rnd:
                res.setdefault((cfg, "score_s1"), []).append(t)
assert int(counts.sum().item()) == R * N
for (cfg, k), ts in sorted(res.items()):
    med = float(np.median(ts))
    print("%-8s %-9s median %.4f ms  min %.4f  -> %.0f GB/s (N bytes/bin)  %.2f Gbins/s" % (
        cfg, k, med, min(ts), R * N / med / 1e6, R / med / 1e6))

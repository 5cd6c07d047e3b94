#!/usr/bin/env python3
"""k_bin_hist time against the bytes of H written per bin (state-model sizes 2..17 on the generic path, same matrix)."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from epilogos_amd import engine
engine.require_gpu()
R, N = 15_000_000, 833
X = engine.alloc_states(R, N); bench.generate_shard(torch, X, N, 18, 0)
for rnd in range(2):
    for S in (17, 16, 14, 12, 10, 8, 6, 4, 2):
        H = torch.empty((R, S), dtype=torch.int16, device="cuda"); counts = torch.zeros(S, dtype=torch.int64, device="cuda")
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            counts.zero_(); e0.record(); engine.bin_hist(X, N, S, counts=counts, H=H); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t0 = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            counts.zero_(); e0.record(); engine.bin_hist(X, N, S, counts=counts, want_hist=False); e1.record(); torch.cuda.synchronize()
            t0.append(e0.elapsed_time(e1))
        print("round %d S=%2d: %2d B/bin written: %.3f ms; no H: %.3f ms" % (rnd, S, 2 * S, float(np.median(ts[1:])), float(np.median(t0[1:]))), flush=True)
        del H

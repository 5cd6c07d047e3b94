#!/usr/bin/env python3
"""k_bin_hist under different physical placements of X and H, for a PMC pass (tools/placement_pmc.sh runs this under
rocprofv3 once per counter group; tools/placement_pmc_digest.py joins counters and durations per dispatch).
Every placement is launched LAUNCHES times with the H store and once without (counts only); the launch order is printed
so the digest can label dispatches: placement p -> LAUNCHES x 'H' then 1 x 'noH'."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

engine.require_gpu()
R, N, S = 15_000_000, 833, 18
LAUNCHES, PLACEMENTS = 3, 8
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
keep = []
X = engine.alloc_states(R, N)
bench.generate_shard(torch, X, N, S, 0)
H = torch.empty((R, S), dtype=torch.int16, device="cuda")
for p in range(PLACEMENTS):
    if p and p % 2 == 1:                 # new H
        del H; torch.cuda.empty_cache()
        keep.append(torch.empty(200_000_000 + p * 7_000_000, dtype=torch.uint8, device="cuda"))
        H = torch.empty((R, S), dtype=torch.int16, device="cuda")
    elif p:                              # new X (same contents)
        Xn = None
        keep.append(torch.empty(300_000_000 + p * 11_000_000, dtype=torch.uint8, device="cuda"))
        Xn = engine.alloc_states(R, N)
        Xn.copy_(X)
        del X; torch.cuda.empty_cache()
        X = Xn
    ts = []
    for k in range(LAUNCHES + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.bin_hist(X, N, S, counts=counts, H=H if k < LAUNCHES else None, want_hist=k < LAUNCHES)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("placement %d: X@%x H@%x  with H %s ms, counts only %.3f ms" % (p, X.data_ptr(), H.data_ptr(), " ".join("%.3f" % t for t in ts[:LAUNCHES]), ts[-1]), flush=True)

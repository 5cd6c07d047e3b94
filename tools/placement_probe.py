#!/usr/bin/env python3
"""Does the time of k_bin_hist depend on where the driver places the buffers physically?  Re-allocates the state matrix
and the histogram buffer independently a few times inside one process and times the kernel with and without its H
store (tuning aid; findings in DESIGN.md)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from epilogos_amd import engine  # noqa: E402

engine.require_gpu()
R, N, S = 15_000_000, 833, 18


def timeit(X, H, counts, n=6):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        counts.zero_(); e0.record(); engine.bin_hist(X, N, S, counts=counts, H=H, want_hist=H is not None); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


counts = torch.zeros(S, dtype=torch.int64, device="cuda")
keep = []
X = engine.alloc_states(R, N); X.fill_(17)
H = torch.empty((R, S), dtype=torch.int16, device="cuda")
for trial in range(10):
    what = "same buffers"
    if trial in (2, 3, 4, 8):            # new H only
        del H; torch.cuda.empty_cache()
        keep.append(torch.empty(200_000_000 + trial * 7_000_000, dtype=torch.uint8, device="cuda"))
        H = torch.empty((R, S), dtype=torch.int16, device="cuda"); what = "new H"
    if trial in (5, 6, 7, 9):            # new X only
        del X; torch.cuda.empty_cache()
        keep.append(torch.empty(300_000_000 + trial * 11_000_000, dtype=torch.uint8, device="cuda"))
        X = engine.alloc_states(R, N); X.fill_(17); what = "new X"
    print("trial %d (%-12s): with H store %.3f ms, counts only %.3f ms" % (trial, what, timeit(X, H, counts), timeit(X, None, counts)), flush=True)

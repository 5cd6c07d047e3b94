#!/usr/bin/env python3
"""k_bin_hist time with H (a) in its own allocation, (b) at the end of the allocation that holds X, (c) in its own
allocation made after the caching allocator has been churned -- to see whether one arena avoids the slow placements."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from epilogos_amd import engine  # noqa: E402

engine.require_gpu()
R, N, S = 15_000_000, 833, 18
ldx = 848
counts = torch.zeros(S, dtype=torch.int64, device="cuda")


def timeit(X, H, n=6):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        counts.zero_(); e0.record(); engine.bin_hist(X, N, S, counts=counts, H=H, want_hist=H is not None); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


for rnd in range(3):
    arena = torch.empty(R * ldx + R * S * 2 + 4096, dtype=torch.int8, device="cuda")
    X = arena[: R * ldx].view(R, ldx)
    X.fill_(17)
    off = (R * ldx + 255) // 256 * 256
    H_in = arena[off: off + R * S * 2].view(torch.int16).view(R, S)
    H_own = torch.empty((R, S), dtype=torch.int16, device="cuda")
    t_in, t_own, t_ro = timeit(X, H_in), timeit(X, H_own), timeit(X, None)
    junk = [torch.empty(int(s), dtype=torch.uint8, device="cuda") for s in np.random.default_rng(rnd).integers(1 << 20, 1 << 28, 40)]
    del junk[::2]
    H_late = torch.empty((R, S), dtype=torch.int16, device="cuda")
    t_late = timeit(X, H_late)
    print("round %d: H inside X's allocation %.3f ms | own allocation %.3f ms | own allocation after churn %.3f ms | no H %.3f ms"
          % (rnd, t_in, t_own, t_late, t_ro), flush=True)
    del arena, X, H_in, H_own, H_late, junk
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""k_bin_hist time against the offset of H inside one large allocation (X fixed): looks for the period of the
read-stream / write-stream interference seen in placement_probe.py."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from epilogos_amd import engine  # noqa: E402

engine.require_gpu()
R, N, S = 15_000_000, 833, 18
counts = torch.zeros(S, dtype=torch.int64, device="cuda")


def timeit(X, H, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        counts.zero_(); e0.record(); engine.bin_hist(X, N, S, counts=counts, H=H); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


X = engine.alloc_states(R, N); X.fill_(17)
size = R * S          # int16 elements
big = torch.empty(size + (1 << 29), dtype=torch.int16, device="cuda")       # 1 GiB of slack (in bytes)
print("X at 0x%x, big at 0x%x" % (X.data_ptr(), big.data_ptr()))
for off_bytes in [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29, (1 << 29) + (1 << 21), 1 << 30]:
    o = off_bytes // 2
    H = big[o:o + size].view(R, S)
    print("H offset %11d B : %.3f ms" % (off_bytes, timeit(X, H)), flush=True)

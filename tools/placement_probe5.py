#!/usr/bin/env python3
"""Is the slow level of k_bin_hist a property of PARTS of the state matrix?  Times the kernel on the whole matrix and on each
eighth of its rows (same process, same buffers).  usage: run a few times; processes land on different levels."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

engine.require_gpu()
R, N, S = 15_000_000, 833, 18
X = engine.alloc_states(R, N)
bench.generate_shard(torch, X, N, S, 0)
H = torch.empty((R, S), dtype=torch.int16, device="cuda")
counts = torch.zeros(S, dtype=torch.int64, device="cuda")


def t(Xs, Hs, n=6):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.bin_hist(Xs, N, S, counts=counts, H=Hs, want_hist=Hs is not None); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


print("whole: with H %.3f ms, counts only %.3f ms   X@%x H@%x" % (t(X, H), t(X, None), X.data_ptr(), H.data_ptr()))
k = R // 8
w = [t(X[i * k:(i + 1) * k], H[i * k:(i + 1) * k]) for i in range(8)]
c = [t(X[i * k:(i + 1) * k], None) for i in range(8)]
print("eighths with H : " + " ".join("%.3f" % v for v in w) + "  sum %.3f" % sum(w))
print("eighths no H   : " + " ".join("%.3f" % v for v in c) + "  sum %.3f" % sum(c))
# the same eighth of X against a different eighth of H (does the pairing matter?)
x0 = X[:k]
print("X eighth 0 with H eighth j: " + " ".join("%.3f" % t(x0, H[j * k:(j + 1) * k]) for j in range(8)))

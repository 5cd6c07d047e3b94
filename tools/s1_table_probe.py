#!/usr/bin/env python3
"""How far is the S1 table the device builds (k_s1_combine: f64 divide + ocml log2) from the one numpy builds on the host
(scores.s1ScoreTable, the reference's own expression)?  ulp distance of the float64 entries, number of float32 entries that differ."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from epilogos_amd import engine
from epilogos_amd.scores import s1ScoreTable

engine.require_gpu()
rng = np.random.default_rng(0)
tot32 = tot = 0
worst = 0
for trial in range(40):
    S = 18
    N = int(rng.choice([833, 379, 342, 10, 127, 2000]))
    c = rng.integers(1, 10**9, S).astype(np.int64)
    if trial % 3 == 0:
        c[rng.integers(0, S)] = 0
    counts = torch.from_numpy(c).cuda()
    q, T64, T32 = engine.s1_tables(counts, N, S)
    torch.cuda.synchronize()
    qh = q.cpu().numpy()
    assert np.array_equal(qh, (c / c.sum()).astype(np.float32))
    t64, t32 = s1ScoreTable(qh, N)
    d64, d32 = T64.cpu().numpy().reshape(N + 1, S), T32.cpu().numpy().reshape(N + 1, S)
    ulp = np.abs(d64.view(np.int64) - t64.view(np.int64))
    nz = (t64 != 0) | (d64 != 0)
    n32 = int((d32.view(np.uint32) != t32.view(np.uint32)).sum())
    tot32 += n32; tot += d32.size
    worst = max(worst, int(ulp[nz].max()) if nz.any() else 0)
    print("N=%4d: f64 entries differing %d of %d (max %d ulp), float32 entries differing %d" % (N, int((ulp[nz] > 0).sum()), int(nz.sum()), int(ulp[nz].max()) if nz.any() else 0, n32))
print("TOTAL float32 entries differing: %d of %d; worst f64 distance %d ulp" % (tot32, tot, worst))

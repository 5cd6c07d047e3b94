#!/usr/bin/env python3
"""Random shapes through both S3 score kernels (k_s3_score_bl, the default, and k_s3_score via EPG_S3_SCORE=bins): the two must
agree to 1e-6 and repeat bit for bit.  GPU box only; tests/test_hip_s3_stress.py is the version that also checks the oracle.
usage: s3_stress.py [--cases 60] [--seed 1]"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from epilogos_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
engine.require_gpu()
rng = np.random.default_rng(a.seed)
worst = 0.0
for case in range(a.cases):
    S = int(rng.integers(2, 21))
    N = int(rng.choice([2, 3, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 129, 200, 257, int(rng.integers(2, 300))]))
    R = int(rng.choice([1, 47, 48, 95, 96, 97, 1439, 1440, 1441, 2879, 2880, 2881, 4321, int(rng.integers(1, 6000))]))
    p = rng.dirichlet(np.full(S, 0.3))
    x = rng.choice(S, size=(R, N), p=p).astype(np.int8)
    if R > 3 and N > 2:
        x[rng.integers(0, R), rng.integers(0, N)] = -1                 # not a state
        x[rng.integers(0, R), rng.integers(0, N)] = 31
    q = rng.random((N, N, S, S)).astype(np.float32) ** 3
    q[rng.random(q.shape) < 0.05] = 0.0                                # masked entries
    q /= q.sum()
    X = engine.states_to_device(x)
    qd = torch.from_numpy(q.reshape(-1)).cuda()
    os.environ.pop("EPG_S3_SCORE", None)
    a32, a64 = engine.score_s3(X, N, S, qd, want32=True, want64=True)
    b32, _ = engine.score_s3(X, N, S, qd, want32=True, want64=False)
    assert torch.equal(a32, b32), ("not reproducible", N, S, R)
    os.environ["EPG_S3_SCORE"] = "bins"
    _, c64 = engine.score_s3(X, N, S, qd, want32=False, want64=True)
    os.environ.pop("EPG_S3_SCORE")
    A, C = a64.cpu().numpy(), c64.cpu().numpy()
    err = np.abs(A - C) / np.maximum(np.abs(C), 1e-9)
    ok = np.allclose(A, C, rtol=1e-6, atol=1e-9)
    worst = max(worst, float((np.abs(A - C) / np.maximum(np.abs(C), 1e-3)).max()))
    if not ok:
        print("MISMATCH N=%d S=%d R=%d max rel %.3g" % (N, S, R, err.max()))
        sys.exit(1)
print("s3 stress: %d shapes, both kernels agree (worst relative difference %.2e), runs repeat bit for bit" % (a.cases, worst))

#!/usr/bin/env python3
"""Does a state-matrix allocation keep its k_bin_hist level?  Three candidates timed uninitialised, then filled with the bench
data, then after the other two are freed and returned to the driver, then after new buffers are allocated."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

N, S, R = 833, 18, 15000000
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
H = torch.empty((R, S), dtype=torch.int16, device="cuda")


def k1(X, H, reps=5):
    engine.bin_hist(X, N, S, counts=counts, H=H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        engine.bin_hist(X, N, S, counts=counts, H=H)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


Xs = [engine.alloc_states(R, N) for _ in range(3)]
print("uninitialised:   " + "  ".join("%.3f" % k1(X, H) for X in Xs), " (sum of first MB: %s)" % [int(X.view(-1)[:1 << 20].to(torch.int64).sum()) for X in Xs])
for X in Xs:
    X.zero_()
print("zero filled:     " + "  ".join("%.3f" % k1(X, H) for X in Xs))
bench.generate_shard(torch, Xs[0], N, S, 0)
for X in Xs[1:]:
    X.copy_(Xs[0])
print("bench data:      " + "  ".join("%.3f" % k1(X, H) for X in Xs))
best = min(range(3), key=lambda i: k1(Xs[i], H))
X = Xs[best]
Xs = None
torch.cuda.empty_cache()
print("kept candidate %d, others returned to the driver: %.3f" % (best, k1(X, H)))
H2 = torch.empty((R, S), dtype=torch.int16, device="cuda")
out32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
print("with a new H, after allocating out32:           %.3f" % k1(X, H2))
del H
torch.cuda.empty_cache()
print("old H freed:                                    %.3f" % k1(X, H2))

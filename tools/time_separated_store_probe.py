#!/usr/bin/env python3
"""GPU box: would separating K1's reads and its histogram store IN TIME remove the memory-class penalty?  Emulation with the kernels
there are: the genome in 15 slices of 1 M bins, per slice a counts-only launch (reads only) followed by a pure write burst of the
slice's 36 MB of histogram rows (torch fill), against k_bin_hist with its interleaved store -- on a PLAIN histogram allocation
(as a rule in the matrix's class) and on the placed one.  The launch gaps of the 30 small launches are measured with an
empty-handed twin (fills of 64 bytes)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from epilogos_amd import engine
engine.require_gpu()
S, N, R = 18, 833, 15_000_000
X = engine.alloc_states(R, N)
bench.generate_shard(torch, X, N, S, 0)
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
Hplain = torch.empty((R, S), dtype=torch.int16, device="cuda")
os.environ["EPILOGOS_PLACEMENT_EAGER"] = "1"
Hplaced = engine.alloc_hist(X, N, S)
print("placement:", engine.placement_report().get("decision"), engine.placement_report().get("ratios"))
tiny = torch.empty(32, dtype=torch.int16, device="cuda")
step = 1_000_000


def timed(fn, reps=5):
    ts = []
    for k in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if k:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def phased(H, real=True):
    for lo in range(0, R, step):
        engine.bin_hist(X[lo:lo + step], N, S, counts=counts, want_hist=False)
        (H[lo:lo + step] if real else tiny).fill_(7)


for name, H in (("plain", Hplain), ("placed", Hplaced)):
    a = timed(lambda: engine.bin_hist(X, N, S, counts=counts, H=H))
    b = timed(lambda: phased(H))
    c = timed(lambda: phased(H, real=False))
    d = timed(lambda: engine.bin_hist(X, N, S, counts=counts, want_hist=False))
    print("%-6s  K1 with the interleaved store %.3f ms | 15 x (counts-only 1 M bins + 36 MB write burst) %.3f ms | the same with 64-byte "
          "fills (launch gaps) %.3f ms | counts only, one launch %.3f ms  -> time-separated estimate %.3f ms" % (name, a, b, c, d, d + (b - c)))

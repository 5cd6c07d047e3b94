#!/usr/bin/env python3
"""k_bin_hist time for every pair of four state-matrix allocations and four histogram allocations (all held at once), then the
whole bench step (K1, combine, score into out32) for the best and the worst pair with two out32 allocations each."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

N, S, R = 833, 18, 15000000
counts = torch.zeros(S, dtype=torch.int64, device="cuda")


def k1(X, H, reps=5):
    engine.bin_hist(X, N, S, counts=counts, H=H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        engine.bin_hist(X, N, S, counts=counts, H=H)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


Xs, Hs = [], []
for i in range(4):
    Xs.append(engine.alloc_states(R, N))
    Hs.append(torch.empty((R, S), dtype=torch.int16, device="cuda"))
bench.generate_shard(torch, Xs[0], N, S, 0)
for X in Xs[1:]:
    X.copy_(Xs[0])
print("X @ " + " ".join("%x" % X.data_ptr() for X in Xs))
print("H @ " + " ".join("%x" % H.data_ptr() for H in Hs))
tab = [[k1(X, H) for H in Hs] for X in Xs]
for i, row in enumerate(tab):
    print("X%d: " % i + "  ".join("%.3f" % v for v in row))
flat = sorted((tab[i][j], i, j) for i in range(4) for j in range(4))
ws = engine.workspace(1, 0, N, S)
q = torch.empty(S, dtype=torch.float32, device="cuda")
for label, (_, i, j) in (("best pair", flat[0]), ("worst pair", flat[-1])):
    for o in range(2):
        out32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tot_k1 = tot = 0.0
        for rep in range(8):
            ev[0].record()
            engine.bin_hist(Xs[i], N, S, counts=counts, H=Hs[j])
            ev[1].record()
            engine.combine_score_s1(counts, Hs[j], N, S, q=q, out32=out32, ws=ws, rezero=True)
            ev[2].record()
            torch.cuda.synchronize()
            if rep >= 2:
                tot_k1 += ev[0].elapsed_time(ev[1]); tot += ev[0].elapsed_time(ev[2])
        print("%s X%d H%d out32@%x: K1 in the step %.3f ms, step %.3f ms" % (label, i, j, out32.data_ptr(), tot_k1 / 6, tot / 6))
        keep = out32

#!/usr/bin/env python3
"""Why is k_bin_hist ~8 % slower inside a step (after the score pass) than in a loop of its own?  The kernel is timed by events
directly after different predecessors: itself, a tiny kernel, a pure writer of the score output's size, a pure reader, the
real score pass, and the score pass followed by a reader that sweeps 512 MB (which pushes the written lines out of the caches).
X and H are put into different memory classes first (tools/placement_map2.py), so only the predecessor varies."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from epilogos_amd import engine  # noqa: E402

N, S = 833, 18
R = int(sys.argv[1]) if len(sys.argv) > 1 else 15_000_000
X = engine.alloc_states(R, N)
X.fill_(17)
X[:, ::7] = 5
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
H, rep = engine.place_hist(X, N, S, tries=9)
print("placement:", rep, flush=True)
out32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
q = torch.full((S,), 1.0 / S, dtype=torch.float32, device="cuda")
ws = engine.workspace(1, 0, N, S)
sweep = torch.empty(512 << 20, dtype=torch.int8, device="cuda")
small = torch.empty(1 << 28, dtype=torch.int8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def score():
    engine.combine_score_s1(counts, H, N, S, q=q, out32=out32, ws=ws, rezero=True)


preds = {
    "k_bin_hist itself": lambda: engine.bin_hist(X, N, S, counts=counts, H=H),
    "tiny kernel": lambda: counts.zero_(),
    "writer of 1.08 GB (fill_ of the score output)": lambda: out32.fill_(0.0),
    "writer of 256 MiB": lambda: small.fill_(0),
    "reader of 0.54 GB (sum over H)": lambda: H.sum(),
    "score pass": score,
    "score pass, then a 512 MiB read sweep": lambda: (score(), sweep.sum()),
    "score pass, then a 512 MiB write sweep": lambda: (score(), sweep.fill_(1)),
}
for name, pred in preds.items():
    ts = []
    for k in range(6):
        engine.bin_hist(X, N, S, counts=counts, H=H)
        pred()
        e0.record()
        engine.bin_hist(X, N, S, counts=counts, H=H)
        e1.record()
        torch.cuda.synchronize()
        counts.zero_()
        if k:
            ts.append(e0.elapsed_time(e1))
    print("%-48s -> k_bin_hist %.3f ms [%.3f .. %.3f]" % (name, sum(ts) / len(ts), min(ts), max(ts)), flush=True)

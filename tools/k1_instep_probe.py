#!/usr/bin/env python3
"""k_bin_hist inside the S1 step against the same launch in a loop of its own: which ingredient of bench.py's loop costs the
~0.2 ms (k1_after_probe.py, constant data and a sync per iteration, sees 0.065 ms)?  Data (constant / the bench's synthetic
states) x queueing (sync after every step / 20 steps queued)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

N, S = 833, 18
R = 15_000_000
X = engine.alloc_states(R, N)
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
out32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
q = torch.empty(S, dtype=torch.float32, device="cuda")
ws = engine.workspace(1, 0, N, S)
H = None
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(20)]


def isolated():
    counts.zero_()
    engine.bin_hist(X, N, S, counts=counts, H=H)
    ev[0][0].record()
    for _ in range(5):
        engine.bin_hist(X, N, S, counts=counts, H=H)
    ev[0][1].record()
    torch.cuda.synchronize()
    return ev[0][0].elapsed_time(ev[0][1]) / 5


def steps(sync, events=True):
    counts.zero_()
    for k in range(20):
        if events or k == 0:
            ev[k][0].record()
        engine.bin_hist(X, N, S, counts=counts, H=H)
        if events:
            ev[k][1].record()
        engine.combine_score_s1(counts, H, N, S, q=q, out32=out32, ws=ws, rezero=True)
        if events or k == 19:
            ev[k][2].record()
        if sync:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if not events:
        return float("nan"), float("nan"), ev[0][0].elapsed_time(ev[19][2]) / 20
    k1 = sum(e[0].elapsed_time(e[1]) for e in ev[2:]) / 18
    rest = sum(e[1].elapsed_time(e[2]) for e in ev[2:]) / 18
    return k1, rest, ev[2][0].elapsed_time(ev[19][2]) / 18


for data in ("constant", "bench"):
    if data == "constant":
        X.fill_(17)
        X[:, ::7] = 5
    else:
        bench.generate_shard(torch, X, N, S, 0)
    if H is None:
        H, rep = engine.place_hist(X, N, S)
        print("placement:", rep, flush=True)
    print("%s data: isolated K1 %.3f ms" % (data, isolated()))
    for sync in (True, False):
        k1, rest, st = steps(sync)
        print("  %-22s K1 %.3f rest %.3f step %.3f" % ("sync after every step:" if sync else "20 steps queued:", k1, rest, st), flush=True)
    k1, rest, st = steps(False, events=False)
    print("  20 steps queued, no events inside: step %.3f" % st)
    print("  isolated again %.3f" % isolated(), flush=True)

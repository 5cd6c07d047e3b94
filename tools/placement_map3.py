#!/usr/bin/env python3
"""Does a pure READ stream get faster when it draws from several memory classes at once (placement_map2.py: three classes,
presumably the ranks of the HBM stacks -- a refresh of one rank would then hide behind reads of the others)?
4 GiB blocks are classified as in placement_map2.py; then k_bin_hist (counts only: no store) runs on THREE blocks at once,
one launch per stream with a third of the persistent grid each, the three blocks taken from one class or from three."""
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from epilogos_amd import _abi, engine  # noqa: E402

N, S = 833, 18
ldx = engine.padded_width(N)
BLOCK = 4 << 30
R = BLOCK // ldx
lib = _abi.load()
lib.epg_debug_set_variant.argtypes = [ctypes.c_int, ctypes.c_int]
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
free, total = torch.cuda.mem_get_info()
nblocks = int((free - (4 << 30)) // BLOCK)
blocks = [torch.empty(BLOCK, dtype=torch.int8, device="cuda") for _ in range(nblocks)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def as_x(b):
    return b[:R * ldx].view(R, ldx)


def t_k1(X, H, reps=3):
    engine.bin_hist(X, N, S, counts=counts, H=H)
    e0.record()
    for _ in range(reps):
        engine.bin_hist(X, N, S, counts=counts, H=H)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def classify_against(j):
    Hj = blocks[j][BLOCK // 2:BLOCK // 2 + R * S * 2].view(torch.int16).view(R, S)
    out = []
    for i, b in enumerate(blocks):
        if i == j:
            out.append(1)
            continue
        t = t_k1(as_x(b), Hj)
        out.append(1 if t > 0.855 else (0 if t < 0.775 else -1))
    return out


c0 = classify_against(0)
j1 = next(i for i, c in enumerate(c0) if c == 0)
c1 = classify_against(j1)
cls = ["A" if (a == 1 and b == 0) else "B" if (a == 0 and b == 1) else "C" if (a == 0 and b == 0) else "?" for a, b in zip(c0, c1)]
print("classes: " + "".join(cls), flush=True)
by = {k: [i for i, c in enumerate(cls) if c == k] for k in "ABC"}

streams = [torch.cuda.Stream() for _ in range(3)]
cs = [torch.zeros(S, dtype=torch.int64, device="cuda") for _ in range(3)]
ends = [torch.cuda.Event(enable_timing=True) for _ in range(3)]


def trio(ids, bpc, reps=4):
    """three counts-only launches at once (one per stream, bpc blocks per CU each) over the blocks `ids`; ms per round"""
    lib.epg_debug_set_variant(0, bpc)
    ts = []
    for k in range(reps + 1):
        torch.cuda.synchronize()
        e0.record()
        for s, c, e, i in zip(streams, cs, ends, ids):
            s.wait_event(e0)
            with torch.cuda.stream(s):
                engine.bin_hist(as_x(blocks[i]), N, S, counts=c, want_hist=False)
                e.record()
        torch.cuda.synchronize()
        if k:
            ts.append(max(e0.elapsed_time(e) for e in ends))
    lib.epg_debug_set_variant(0, 4)
    return sum(ts) / len(ts), min(ts)


for ids in (by["A"][:1], by["B"][:1], by["C"][:1]):
    print("one block alone, counts only: %.3f ms" % t_k1(as_x(blocks[ids[0]]), None), flush=True)
for bpc in (1, 2):
    print("three launches at once, %d block(s) per CU each (12 GiB read per round):" % bpc)
    for name, ids in (("A A A", by["A"][:3]), ("B B B", by["B"][:3]), ("C C C", by["C"][:3]),
                      ("A B C", [by["A"][0], by["B"][0], by["C"][0]]), ("A B C'", [by["A"][1], by["B"][1], by["C"][1]]),
                      ("A A B", by["A"][:2] + by["B"][:1]), ("B C C", by["B"][:1] + by["C"][:2])):
        if len(ids) == 3:
            mean, best = trio(ids, bpc)
            print("  %-7s blocks %-12s mean %.3f ms  best %.3f ms  -> %.0f GB/s" % (name, ids, mean, best, 3 * R * N / best / 1e6), flush=True)

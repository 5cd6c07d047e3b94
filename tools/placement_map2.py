#!/usr/bin/env python3
"""Second map of the device-memory classes (after placement_map.py): 4 GiB blocks over the whole device memory are sorted into
classes by timing k_bin_hist (matrix in block i, histogram in block j: same class = slow), then three questions:
  1. does a plain device-to-device copy (torch copy_) see the classes (a cheap classifier)?
  2. does the score pass (reads H, writes the float32 scores) see them?
  3. in a whole S1 step (K1 -> combine -> score), where should the SCORE OUTPUT go: the class of X, of H, or the third?"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from epilogos_amd import engine  # noqa: E402

N, S = 833, 18
ldx = engine.padded_width(N)
BLOCK = 4 << 30
R = BLOCK // ldx
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
free, total = torch.cuda.mem_get_info()
nblocks = int((free - (4 << 30)) // BLOCK)
blocks = [torch.empty(BLOCK, dtype=torch.int8, device="cuda") for _ in range(nblocks)]
print("%d blocks of 4 GiB held, %d bins per block" % (nblocks, R), flush=True)
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))


def as_x(b):
    return b[:R * ldx].view(R, ldx)


def as_h(b, off=0):
    return b[off:off + R * S * 2].view(torch.int16).view(R, S)


def as_out(b, off=0):
    return b[off:off + R * S * 4].view(torch.float32).view(R, S)


def t_k1(X, H, reps=3):
    engine.bin_hist(X, N, S, counts=counts, H=H)
    e0.record()
    for _ in range(reps):
        engine.bin_hist(X, N, S, counts=counts, H=H)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def classify_against(j):
    """1 where block i is in block j's class (k_bin_hist slow), 0 where not, -1 for in-between (a block over a boundary)."""
    Hj = as_h(blocks[j], off=BLOCK // 2)
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
cls = []
for a, b in zip(c0, c1):
    cls.append("A" if (a == 1 and b == 0) else "B" if (a == 0 and b == 1) else "C" if (a == 0 and b == 0) else "?")
print("classes (A = block 0's, B = block %d's, C = neither, ? = mixed):\n%s" % (j1, "".join(cls)), flush=True)
by = {k: [i for i, c in enumerate(cls) if c == k] for k in "ABC"}
print({k: len(v) for k, v in by.items()}, flush=True)

# 1. copy probe: 1 GiB from block i to block j
CB = 1 << 30


def t_copy(i, j, reps=3):
    src, dst = blocks[i][:CB], blocks[j][BLOCK // 2:BLOCK // 2 + CB]
    dst.copy_(src)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def pairs(ka, kb, n=6):
    out = []
    for i in by[ka][:n]:
        for j in by[kb][:n]:
            if i != j:
                out.append((i, j))
    return out[:12]


print("\n1. copy of 1 GiB, ms (src class -> dst class): mean [min .. max]")
for ka in "ABC":
    for kb in "ABC":
        ts = [t_copy(i, j) for i, j in pairs(ka, kb)]
        if ts:
            print("  %s -> %s: %.3f [%.3f .. %.3f]" % (ka, kb, sum(ts) / len(ts), min(ts), max(ts)), flush=True)

# 2. score pass: H in class ka, out32 in class kb
q = torch.full((S,), 1.0 / S, dtype=torch.float32, device="cuda")
ws = engine.workspace(1, 0, N, S, device="cuda")
X0 = as_x(blocks[by["A"][0]])
X0.fill_(17)
X0[:, ::7] = 5


def t_score(i, j, reps=3):
    H, o = as_h(blocks[i]), as_out(blocks[j], off=BLOCK // 2)
    engine.bin_hist(X0, N, S, counts=counts, H=H)
    engine.score_s1_from_binhist(H, N, S, q, out32=o, ws=ws)
    e0.record()
    for _ in range(reps):
        engine.score_s1_from_binhist(H, N, S, q, out32=o, ws=ws)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("\n2. score pass over %d bins, ms (H class -> scores class)" % R)
for ka in "ABC":
    for kb in "ABC":
        ts = [t_score(i, j) for i, j in pairs(ka, kb, 4)[:6] if i != by["A"][0]]
        if ts:
            print("  %s -> %s: %.3f [%.3f .. %.3f]" % (ka, kb, sum(ts) / len(ts), min(ts), max(ts)), flush=True)

# 3. the whole step with X in A: H in {A,B}, scores in {A,B,C}
print("\n3. whole S1 step over %d bins, X in class A: (H class, scores class) -> K1 ms in the step, rest ms, step ms" % R)
xa = by["A"][0]
for kh in "ABC":
    for ko in "ABC":
        res = []
        for ih in [i for i in by[kh] if i != xa][:2]:
            for io in [i for i in by[ko] if i not in (xa, ih)][:2]:
                H, o = as_h(blocks[ih]), as_out(blocks[io], off=BLOCK // 2)
                k1s, rests = [], []
                counts.zero_()
                for k in range(5):
                    e0.record()
                    engine.bin_hist(X0, N, S, counts=counts, H=H)
                    e1.record()
                    engine.combine_score_s1(counts, H, N, S, q=q, out32=o, ws=ws, rezero=True)
                    e2.record()
                    torch.cuda.synchronize()
                    if k:
                        k1s.append(e0.elapsed_time(e1))
                        rests.append(e1.elapsed_time(e2))
                # back to back (what bench.py times)
                e0.record()
                for k in range(4):
                    engine.bin_hist(X0, N, S, counts=counts, H=H)
                    engine.combine_score_s1(counts, H, N, S, q=q, out32=o, ws=ws, rezero=True)
                e1.record()
                torch.cuda.synchronize()
                res.append((sum(k1s) / 4, sum(rests) / 4, e0.elapsed_time(e1) / 4))
        if res:
            m = [sum(r[k] for r in res) / len(res) for k in range(3)]
            print("  H %s scores %s: K1 %.3f rest %.3f | back-to-back step %.3f" % (kh, ko, m[0], m[1], m[2]), flush=True)

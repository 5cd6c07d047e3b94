#!/usr/bin/env python3
"""k_bin_hist on the SAME state matrix placed at different byte offsets inside ONE allocation (the physical pages stay, the
matrix's alignment against them moves), then in fresh allocations: does the launch time follow the offset or the pages?"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

N, S, R = 833, 18, 15000000
ldx = engine.padded_width(N)
nbytes = R * ldx
master = torch.empty((R, ldx), dtype=torch.int8, device="cuda")
bench.generate_shard(torch, master, N, S, 0)
H = torch.empty((R, S), dtype=torch.int16, device="cuda")
counts = torch.zeros(S, dtype=torch.int64, device="cuda")


def t(X, reps=10):
    engine.bin_hist(X, N, S, counts=counts, H=H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        engine.bin_hist(X, N, S, counts=counts, H=H)
    e1.record()
    torch.cuda.synchronize()
    with_h = e0.elapsed_time(e1) / reps
    e0.record()
    for _ in range(reps):
        engine.bin_hist(X, N, S, counts=counts, want_hist=False)
    e1.record()
    torch.cuda.synchronize()
    return with_h, e0.elapsed_time(e1) / reps


print("master (its own allocation): with H %.3f ms, counts only %.3f ms" % t(master))
slack = 96 << 20
for round_ in range(3):
    arena = torch.empty(nbytes + slack, dtype=torch.int8, device="cuda")
    base = arena.data_ptr()
    out = []
    for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 5 << 20, 16 << 20, 33 << 20, 64 << 20):
        X = arena[off:off + nbytes].view(R, ldx)
        X.copy_(master)
        w, c = t(X)
        out.append("%s: %.3f (%.3f)" % (("%d K" % (off >> 10)) if off < (1 << 20) else ("%d M" % (off >> 20)), w, c))
    print("arena %d @%x:  " % (round_, base) + "  ".join(out), flush=True)
    keep = arena if round_ == 0 else None        # hold the first arena so that the next ones get other pages
    del arena

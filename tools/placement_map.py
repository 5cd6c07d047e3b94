#!/usr/bin/env python3
"""A map of k_bin_hist's level over the whole device memory: 4 GiB blocks allocated one after the other until ~270 GB are held,
the kernel (with the histogram store into one fixed buffer, and without) timed on every block.  Blocks come out of the driver's
allocator in order, so the index is a proxy for the physical position."""
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
H = torch.empty((R, S), dtype=torch.int16, device="cuda")
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
free, total = torch.cuda.mem_get_info()
nblocks = int((free - (6 << 30)) // BLOCK)
blocks = []
for i in range(nblocks):
    blocks.append(torch.empty(BLOCK, dtype=torch.int8, device="cuda"))
print("%d blocks of 4 GiB held (%.0f of %.0f GB), %d bins per block" % (nblocks, nblocks * 4.29, total / 1e9, R))


def t(X, with_h, reps=4):
    engine.bin_hist(X, N, S, counts=counts, H=H if with_h else None, want_hist=with_h)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        engine.bin_hist(X, N, S, counts=counts, H=H if with_h else None, want_hist=with_h)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


H0 = H
row_c = [t(b[:R * ldx].view(R, ldx), False) for b in blocks]
print("counts only          : " + " ".join("%.2f" % v for v in row_c), flush=True)
# the histogram buffer in its own (first) allocation, then INSIDE one of the blocks (the block itself is skipped: '....')
for where in (None, 3, 10, 21, 24, 40, 50, 60, 64, 68):
    if where is None:
        H = H0
    else:
        if where >= len(blocks):
            continue
        H = blocks[where][:R * S * 2].view(torch.int16).view(R, S)
    row = []
    for i, b in enumerate(blocks):
        row.append("...." if i == where else "%.2f" % t(b[:R * ldx].view(R, ldx), True))
    print("H in %-16s: " % ("its own allocation" if where is None else "block %d" % where) + " ".join(row), flush=True)

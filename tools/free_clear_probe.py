#!/usr/bin/env python3
"""Does returning a lot of device memory to the driver slow kernels down for a while (the driver wipes freed VRAM in the
background)?  k_bin_hist (counts only: a pure read stream) is timed every ~20 ms before and after 160 GiB are freed."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from epilogos_amd import engine  # noqa: E402

N, S = 833, 18
R = 15_000_000
X = engine.alloc_states(R, N)
X.fill_(17)
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t():
    e0.record()
    engine.bin_hist(X, N, S, counts=counts, want_hist=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    t()
print("before: " + " ".join("%.3f" % t() for _ in range(8)), flush=True)
GB = int(sys.argv[1]) if len(sys.argv) > 1 else 160
big = [torch.empty(8 << 30, dtype=torch.int8, device="cuda") for _ in range(GB // 8)]
print("holding %d GiB: " % GB + " ".join("%.3f" % t() for _ in range(8)), flush=True)
t0 = time.perf_counter()
del big
torch.cuda.empty_cache()
t_free = time.perf_counter() - t0
print("freed in %.3f s" % t_free, flush=True)
out = []
while time.perf_counter() - t0 < 4.0:
    out.append((time.perf_counter() - t0, t()))
    time.sleep(0.02)
print("after (s since free: ms): " + " ".join("%.2f:%.3f" % v for v in out), flush=True)

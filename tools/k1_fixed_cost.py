import sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from epilogos_amd import engine, _abi
N, S = 833, 18
X = engine.alloc_states(15_000_000, N)
bench.generate_shard(torch, X, N, S, 0)
lib = _abi.load()
lib.epg_test_force.argtypes = [ctypes.c_int32, ctypes.c_int32]
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
def t(R, H, reps=30):
    Xs = X[:R]
    for _ in range(3): engine.bin_hist(Xs, N, S, counts=counts, H=H, want_hist=H is not None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): engine.bin_hist(Xs, N, S, counts=counts, H=H, want_hist=H is not None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
Hfull = engine.alloc_hist(X, N, S)
print(engine.placement_report())
for bpc in (2, 4, 3):
    lib.epg_test_force(4, bpc)
    for R in (468750, 937500, 1875000, 3750000, 7500000, 15000000):
        a = t(R, Hfull[:R]); b = t(R, None)
        print("blocks/CU %d  R %8d  K1 %.4f ms (%.3f of spec)  counts-only %.4f ms" % (bpc, R, a, R * N / a / 1e6 / 8000, b), flush=True)

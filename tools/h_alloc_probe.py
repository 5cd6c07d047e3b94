#!/usr/bin/env python3
"""k_bin_hist time against HOW the histogram buffer H (and the state matrix X) is allocated: torch's caching allocator, plain
hipMalloc, hipExtMallocWithFlags fine-grained / uncached / physically contiguous.  Question: is the placement-dependent cost of
the H store a property of the memory type or of where the pages happen to sit?  Calls the C ABI with raw
device pointers."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from epilogos_amd import _abi, engine  # noqa: E402

engine.require_gpu()
R, N, S = 15_000_000, 833, 18
ldx = engine.padded_width(N)
hip = C.CDLL(torch.__path__[0] + "/lib/libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
FLAGS = {"default": 0x0, "finegrained": 0x1, "uncached": 0x3, "contiguous": 0x4}


def alloc(nbytes, kind):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), nbytes) if kind == "hipMalloc" else hip.hipExtMallocWithFlags(C.byref(p), nbytes, FLAGS[kind])
    return p if rc == 0 and p.value else None


Xt = engine.alloc_states(R, N)
bench.generate_shard(torch, Xt, N, S, 0)
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(xp, hp, n=7):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _abi.call("epg_bin_hist", xp, R, N, ldx, S, hp, C.c_void_p(counts.data_ptr()), st)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


xp_torch = C.c_void_p(Xt.data_ptr())
print("X torch, counts only      : %.3f ms" % timeit(xp_torch, None), flush=True)
for rnd in range(3):
    Ht = torch.empty((R, S), dtype=torch.int16, device="cuda")
    print("round %d: X torch, H torch  : %.3f ms" % (rnd, timeit(xp_torch, C.c_void_p(Ht.data_ptr()))), flush=True)
    for kind in ("hipMalloc", "finegrained", "uncached", "contiguous"):
        hp = alloc(R * S * 2, kind)
        if hp is None:
            print("         H %-12s: allocation failed" % kind, flush=True)
            continue
        print("         X torch, H %-12s: %.3f ms" % (kind, timeit(xp_torch, hp)), flush=True)
        hip.hipFree(hp)
    del Ht
    torch.cuda.empty_cache()
# X itself physically contiguous / plain hipMalloc
for kind in ("hipMalloc", "contiguous"):
    xp = alloc(R * ldx, kind)
    if xp is None:
        print("X %s: allocation failed" % kind)
        continue
    hip.hipMemcpy(xp, C.c_void_p(Xt.data_ptr()), R * ldx, 3)            # device to device
    for hk in ("hipMalloc", "contiguous", "uncached"):
        hp = alloc(R * S * 2, hk)
        if hp is None:
            continue
        print("X %-10s, H %-12s: %.3f ms   (counts only %.3f ms)" % (kind, hk, timeit(xp, hp), timeit(xp, None)), flush=True)
        hip.hipFree(hp)
    hip.hipFree(xp)

#!/usr/bin/env python3
"""k_bin_hist on a state matrix mapped through the HIP virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap) with
1 GiB-aligned virtual addresses and one physical handle, against torch allocations in the same process: does an explicit
mapping pin the fast level?"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import _abi, engine  # noqa: E402

hip = C.CDLL("libamdhip64.so")
N, S, R = 833, 18, 15000000
ldx = engine.padded_width(N)
nbytes = R * ldx


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p),
                ("allocFlags", C.c_ubyte * 4)]     # compressionType, gpuDirectRDMACapable, usage (u16)


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (what, rc))


def vmm_alloc(size, va_align):
    prop = Prop()
    prop.type = 1                      # hipMemAllocationTypePinned
    prop.location.type = 1             # hipMemLocationTypeDevice
    prop.location.id = 0
    gran = C.c_size_t()
    chk(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), "granularity")   # 1 = recommended
    size = (size + gran.value - 1) // gran.value * gran.value
    handle = C.c_void_p()
    chk(hip.hipMemCreate(C.byref(handle), C.c_size_t(size), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
    ptr = C.c_void_p()
    chk(hip.hipMemAddressReserve(C.byref(ptr), C.c_size_t(size), C.c_size_t(va_align), C.c_void_p(0), C.c_ulonglong(0)), "reserve")
    chk(hip.hipMemMap(ptr, C.c_size_t(size), C.c_size_t(0), handle, C.c_ulonglong(0)), "hipMemMap")
    acc = Access()
    acc.location.type = 1
    acc.location.id = 0
    acc.flags = 3                      # read + write
    chk(hip.hipMemSetAccess(ptr, C.c_size_t(size), C.byref(acc), C.c_size_t(1)), "set access")
    return ptr.value, size, gran.value


torch.zeros(1, device="cuda")
master = engine.alloc_states(R, N)
bench.generate_shard(torch, master, N, S, 0)
H = torch.empty((R, S), dtype=torch.int16, device="cuda")
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def k1_ptr(xptr, with_h, reps=5):
    def go():
        _abi.call("epg_bin_hist", xptr, R, N, ldx, S, H.data_ptr() if with_h else None, counts.data_ptr(), st)
    go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("torch allocation      @%x: with H %.3f ms, counts only %.3f ms" % (master.data_ptr(), k1_ptr(master.data_ptr(), True), k1_ptr(master.data_ptr(), False)))
for align in (2 << 20, 1 << 30, 1 << 30, 2 << 20):
    ptr, size, gran = vmm_alloc(nbytes, align)
    chk(hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(master.data_ptr()), C.c_size_t(nbytes), 3), "copy")   # device to device
    torch.cuda.synchronize()
    print("VMM mapping (granularity %d MiB, VA aligned %4d MiB) @%x: with H %.3f ms, counts only %.3f ms"
          % (gran >> 20, align >> 20, ptr, k1_ptr(ptr, True), k1_ptr(ptr, False)), flush=True)
X2 = engine.alloc_states(R, N)
X2.copy_(master)
print("second torch allocation @%x: with H %.3f ms, counts only %.3f ms" % (X2.data_ptr(), k1_ptr(X2.data_ptr(), True), k1_ptr(X2.data_ptr(), False)))

#!/usr/bin/env python3
"""k_bin_hist against the PHYSICAL placement of its buffers: the eight probes of round 2 (formerly placement_probe.py,
placement_probe2.py .. placement_probe9.py; DESIGN.md 3 quotes their results), one script.  GPU box only.

    python tools/placement_probe.py --what realloc|offset|arena|eighths|inalloc|pairs|persist|vmm

Each probe is the body of its former script, unchanged (run one probe per process: they allocate most of the device).
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def probe_realloc():
    """Does the time of k_bin_hist depend on where the driver places the buffers physically?  Re-allocates the state matrix
    and the histogram buffer independently a few times inside one process and times the kernel with and without its H
    store (tuning aid; findings in DESIGN.md)."""
    import sys
    from pathlib import Path

    import numpy as np
    import torch

    from epilogos_amd import engine  # noqa: E402

    engine.require_gpu()
    R, N, S = 15_000_000, 833, 18


    def timeit(X, H, counts, n=6):
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            counts.zero_(); e0.record(); engine.bin_hist(X, N, S, counts=counts, H=H, want_hist=H is not None); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts[1:]))


    counts = torch.zeros(S, dtype=torch.int64, device="cuda")
    keep = []
    X = engine.alloc_states(R, N); X.fill_(17)
    H = torch.empty((R, S), dtype=torch.int16, device="cuda")
    for trial in range(10):
        what = "same buffers"
        if trial in (2, 3, 4, 8):            # new H only
            del H; torch.cuda.empty_cache()
            keep.append(torch.empty(200_000_000 + trial * 7_000_000, dtype=torch.uint8, device="cuda"))
            H = torch.empty((R, S), dtype=torch.int16, device="cuda"); what = "new H"
        if trial in (5, 6, 7, 9):            # new X only
            del X; torch.cuda.empty_cache()
            keep.append(torch.empty(300_000_000 + trial * 11_000_000, dtype=torch.uint8, device="cuda"))
            X = engine.alloc_states(R, N); X.fill_(17); what = "new X"
        print("trial %d (%-12s): with H store %.3f ms, counts only %.3f ms" % (trial, what, timeit(X, H, counts), timeit(X, None, counts)), flush=True)


def probe_offset():
    """k_bin_hist time against the offset of H inside one large allocation (X fixed): looks for the period of the
    read-stream / write-stream interference seen in placement_probe.py."""
    import sys
    from pathlib import Path

    import numpy as np
    import torch

    from epilogos_amd import engine  # noqa: E402

    engine.require_gpu()
    R, N, S = 15_000_000, 833, 18
    counts = torch.zeros(S, dtype=torch.int64, device="cuda")


    def timeit(X, H, n=5):
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            counts.zero_(); e0.record(); engine.bin_hist(X, N, S, counts=counts, H=H); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts[1:]))


    X = engine.alloc_states(R, N); X.fill_(17)
    size = R * S          # int16 elements
    big = torch.empty(size + (1 << 29), dtype=torch.int16, device="cuda")       # 1 GiB of slack (in bytes)
    print("X at 0x%x, big at 0x%x" % (X.data_ptr(), big.data_ptr()))
    for off_bytes in [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 1 << 21, 3 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29, (1 << 29) + (1 << 21), 1 << 30]:
        o = off_bytes // 2
        H = big[o:o + size].view(R, S)
        print("H offset %11d B : %.3f ms" % (off_bytes, timeit(X, H)), flush=True)


def probe_arena():
    """k_bin_hist time with H (a) in its own allocation, (b) at the end of the allocation that holds X, (c) in its own
    allocation made after the caching allocator has been churned -- to see whether one arena avoids the slow placements."""
    import sys
    from pathlib import Path

    import numpy as np
    import torch

    from epilogos_amd import engine  # noqa: E402

    engine.require_gpu()
    R, N, S = 15_000_000, 833, 18
    ldx = 848
    counts = torch.zeros(S, dtype=torch.int64, device="cuda")


    def timeit(X, H, n=6):
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            counts.zero_(); e0.record(); engine.bin_hist(X, N, S, counts=counts, H=H, want_hist=H is not None); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts[1:]))


    for rnd in range(3):
        arena = torch.empty(R * ldx + R * S * 2 + 4096, dtype=torch.int8, device="cuda")
        X = arena[: R * ldx].view(R, ldx)
        X.fill_(17)
        off = (R * ldx + 255) // 256 * 256
        H_in = arena[off: off + R * S * 2].view(torch.int16).view(R, S)
        H_own = torch.empty((R, S), dtype=torch.int16, device="cuda")
        t_in, t_own, t_ro = timeit(X, H_in), timeit(X, H_own), timeit(X, None)
        junk = [torch.empty(int(s), dtype=torch.uint8, device="cuda") for s in np.random.default_rng(rnd).integers(1 << 20, 1 << 28, 40)]
        del junk[::2]
        H_late = torch.empty((R, S), dtype=torch.int16, device="cuda")
        t_late = timeit(X, H_late)
        print("round %d: H inside X's allocation %.3f ms | own allocation %.3f ms | own allocation after churn %.3f ms | no H %.3f ms"
              % (rnd, t_in, t_own, t_late, t_ro), flush=True)
        del arena, X, H_in, H_own, H_late, junk
        torch.cuda.empty_cache()


def probe_eighths():
    """Is the slow level of k_bin_hist a property of PARTS of the state matrix?  Times the kernel on the whole matrix and on each
    eighth of its rows (same process, same buffers).  usage: run a few times; processes land on different levels."""
    import sys
    from pathlib import Path

    import numpy as np
    import torch

    import bench  # noqa: E402
    from epilogos_amd import engine  # noqa: E402

    engine.require_gpu()
    R, N, S = 15_000_000, 833, 18
    X = engine.alloc_states(R, N)
    bench.generate_shard(torch, X, N, S, 0)
    H = torch.empty((R, S), dtype=torch.int16, device="cuda")
    counts = torch.zeros(S, dtype=torch.int64, device="cuda")


    def t(Xs, Hs, n=6):
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); engine.bin_hist(Xs, N, S, counts=counts, H=Hs, want_hist=Hs is not None); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts[1:]))


    print("whole: with H %.3f ms, counts only %.3f ms   X@%x H@%x" % (t(X, H), t(X, None), X.data_ptr(), H.data_ptr()))
    k = R // 8
    w = [t(X[i * k:(i + 1) * k], H[i * k:(i + 1) * k]) for i in range(8)]
    c = [t(X[i * k:(i + 1) * k], None) for i in range(8)]
    print("eighths with H : " + " ".join("%.3f" % v for v in w) + "  sum %.3f" % sum(w))
    print("eighths no H   : " + " ".join("%.3f" % v for v in c) + "  sum %.3f" % sum(c))
    # the same eighth of X against a different eighth of H (does the pairing matter?)
    x0 = X[:k]
    print("X eighth 0 with H eighth j: " + " ".join("%.3f" % t(x0, H[j * k:(j + 1) * k]) for j in range(8)))


def probe_inalloc():
    """k_bin_hist on the SAME state matrix placed at different byte offsets inside ONE allocation (the physical pages stay, the
    matrix's alignment against them moves), then in fresh allocations: does the launch time follow the offset or the pages?"""
    import sys
    from pathlib import Path

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


def probe_pairs():
    """k_bin_hist time for every pair of four state-matrix allocations and four histogram allocations (all held at once), then the
    whole bench step (K1, combine, score into out32) for the best and the worst pair with two out32 allocations each."""
    import sys
    from pathlib import Path

    import torch  # noqa: E402
    import bench  # noqa: E402
    from epilogos_amd import engine  # noqa: E402

    N, S, R = 833, 18, 15000000
    counts = torch.zeros(S, dtype=torch.int64, device="cuda")


    def k1(X, H, reps=5):
        engine.bin_hist(X, N, S, counts=counts, H=H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            engine.bin_hist(X, N, S, counts=counts, H=H)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps


    Xs, Hs = [], []
    for i in range(4):
        Xs.append(engine.alloc_states(R, N))
        Hs.append(torch.empty((R, S), dtype=torch.int16, device="cuda"))
    bench.generate_shard(torch, Xs[0], N, S, 0)
    for X in Xs[1:]:
        X.copy_(Xs[0])
    print("X @ " + " ".join("%x" % X.data_ptr() for X in Xs))
    print("H @ " + " ".join("%x" % H.data_ptr() for H in Hs))
    tab = [[k1(X, H) for H in Hs] for X in Xs]
    for i, row in enumerate(tab):
        print("X%d: " % i + "  ".join("%.3f" % v for v in row))
    flat = sorted((tab[i][j], i, j) for i in range(4) for j in range(4))
    ws = engine.workspace(1, 0, N, S)
    q = torch.empty(S, dtype=torch.float32, device="cuda")
    for label, (_, i, j) in (("best pair", flat[0]), ("worst pair", flat[-1])):
        for o in range(2):
            out32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tot_k1 = tot = 0.0
            for rep in range(8):
                ev[0].record()
                engine.bin_hist(Xs[i], N, S, counts=counts, H=Hs[j])
                ev[1].record()
                engine.combine_score_s1(counts, Hs[j], N, S, q=q, out32=out32, ws=ws, rezero=True)
                ev[2].record()
                torch.cuda.synchronize()
                if rep >= 2:
                    tot_k1 += ev[0].elapsed_time(ev[1]); tot += ev[0].elapsed_time(ev[2])
            print("%s X%d H%d out32@%x: K1 in the step %.3f ms, step %.3f ms" % (label, i, j, out32.data_ptr(), tot_k1 / 6, tot / 6))
            keep = out32


def probe_persist():
    """Does a state-matrix allocation keep its k_bin_hist level?  Three candidates timed uninitialised, then filled with the bench
    data, then after the other two are freed and returned to the driver, then after new buffers are allocated."""
    import sys
    from pathlib import Path

    import torch  # noqa: E402
    import bench  # noqa: E402
    from epilogos_amd import engine  # noqa: E402

    N, S, R = 833, 18, 15000000
    counts = torch.zeros(S, dtype=torch.int64, device="cuda")
    H = torch.empty((R, S), dtype=torch.int16, device="cuda")


    def k1(X, H, reps=5):
        engine.bin_hist(X, N, S, counts=counts, H=H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            engine.bin_hist(X, N, S, counts=counts, H=H)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps


    Xs = [engine.alloc_states(R, N) for _ in range(3)]
    print("uninitialised:   " + "  ".join("%.3f" % k1(X, H) for X in Xs), " (sum of first MB: %s)" % [int(X.view(-1)[:1 << 20].to(torch.int64).sum()) for X in Xs])
    for X in Xs:
        X.zero_()
    print("zero filled:     " + "  ".join("%.3f" % k1(X, H) for X in Xs))
    bench.generate_shard(torch, Xs[0], N, S, 0)
    for X in Xs[1:]:
        X.copy_(Xs[0])
    print("bench data:      " + "  ".join("%.3f" % k1(X, H) for X in Xs))
    best = min(range(3), key=lambda i: k1(Xs[i], H))
    X = Xs[best]
    Xs = None
    torch.cuda.empty_cache()
    print("kept candidate %d, others returned to the driver: %.3f" % (best, k1(X, H)))
    H2 = torch.empty((R, S), dtype=torch.int16, device="cuda")
    out32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
    print("with a new H, after allocating out32:           %.3f" % k1(X, H2))
    del H
    torch.cuda.empty_cache()
    print("old H freed:                                    %.3f" % k1(X, H2))


def probe_vmm():
    """k_bin_hist on a state matrix mapped through the HIP virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap) with
    1 GiB-aligned virtual addresses and one physical handle, against torch allocations in the same process: does an explicit
    mapping pin the fast level?"""
    import ctypes as C
    import sys
    from pathlib import Path

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


PROBES = {"realloc": probe_realloc, "offset": probe_offset, "arena": probe_arena, "eighths": probe_eighths, "inalloc": probe_inalloc, "pairs": probe_pairs, "persist": probe_persist, "vmm": probe_vmm}


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--what", required=True, choices=sorted(PROBES))
    a, rest = ap.parse_known_args()
    sys.argv = [sys.argv[0]] + rest                    # a probe with arguments of its own parses the rest
    PROBES[a.what]()

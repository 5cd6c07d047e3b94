"""Device plumbing: torch tensors (HBM buffers, streams) in, C-ABI calls out.  PyTorch is used only for
memory, streams and torch.distributed; every arithmetic step is a kernel of libepilogos_hip.so.

State matrices are int8 [R, ldx] with ldx = N rounded up to 16 bytes (pad bytes are never read as states), which
is the layout the streaming kernel's 16-byte loads want.  All functions enqueue on torch's current stream and
return device tensors without synchronising.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._abi import EpilogosHipError

ROW_ALIGN = 16


def require_gpu():
    if not torch.cuda.is_available():
        raise EpilogosHipError(-3, "no HIP device visible to torch: the epilogos_amd engine has no CPU fallback")
    _abi.load()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def padded_width(N):
    return (N + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN


def alloc_states(R, N, device="cuda"):
    """Uninitialised int8 [R, ldx] state matrix; use [:, :N]."""
    return torch.empty((R, padded_width(N)), dtype=torch.int8, device=device)


# ---- where the histogram cache of a RESIDENT matrix goes
PLACE_MIN_BYTES = 1 << 30           # a matrix under 1 GiB is counted in < 0.2 ms: not worth a probe
PLACE_BLOCK = 4 << 30               # candidates are the heads of blocks of this size (see alloc_hist)
PLACE_SEARCH_AT = 2                 # the n-th job on the same resident matrix runs the QUICK search ...
PLACE_TRIES = 8                     # ... at most 8 blocks (32 GiB: about what a process gets from the driver's pool of cleared memory at
PLACE_BUDGET_MS = 50.0              #     0.3 ms per block) and 50 ms of probe time
PLACE_DEEP_AT = 4                   # the n-th job, if the quick search kept the plain allocation, runs the DEEP search, once ...
PLACE_DEEP_TRIES = 24               # ... 24 blocks = 96 GiB, past what one memory class can span; memory beyond the cleared pool costs the
PLACE_DEEP_WALL_MS = 5000.0         #     driver ~30 ms per GiB to hand out (tools/alloc_probe.py): seconds, which only a process that keeps
                                    #     running jobs on the matrix gets back (0.3 ms per 15 M-bin job) -- hence the job count in front of it
PLACE_GAP = 0.03                    # two levels of the probe's ratio are "two memory classes" when they lie >= 3 % apart
PLACE_SURE = 1.10                   # K1 with the H store / K1 counts only at or under this: the store lands in another class, no
                                    # contrast needed (every block ever measured in the matrix's own class: >= 1.136; in another: 1.05-1.105)
PLACE_WIN = 0.01                    # the pick must beat the plain allocation by this much over the WHOLE matrix to replace it
_placement = {}                     # device index -> {"key", "home", "report", "stream", "seen", "tier"}


def _storage_users(t):
    """Tensors (views included) alive on t's storage, or None when this torch cannot tell (placement is then off)."""
    f = getattr(torch._C, "_storage_Use_Count", None)
    if f is None:
        return None
    return int(f(t.untyped_storage()._cdata)) - 1            # (the wrapper made by untyped_storage() counts as one)


def _probe_slices(R, rows=1 << 20):
    """Three row ranges of a matrix (head, middle, tail; a fifth of a genome in all) the class probe runs on: a matrix may
    straddle two memory classes, a histogram cache is good only if it is in neither.  Starts are multiples of 32 bins (whole
    super-tiles, 16-byte aligned histogram rows)."""
    m = min(rows, R // 3) // 32 * 32
    if m <= 0:
        return [(0, R)]
    return [(0, m), ((R - m) // 2 // 32 * 32, (R - m) // 2 // 32 * 32 + m), ((R - m) // 32 * 32, (R - m) // 32 * 32 + m)]


def _probe_ms(X, N, S, Hflat, counts, slices, reps=2):
    """Device time of k_bin_hist over `slices` of X, histogram rows into the same rows of the candidate `Hflat` (None = counts
    only); one untimed pass first."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    H = None if Hflat is None else Hflat[:X.shape[0] * S * 2].view(torch.int16).view(X.shape[0], S)
    for k in range(reps + 1):
        if k == 1:
            ev[0].record()
        for lo, hi in slices:
            bin_hist(X[lo:hi], N, S, counts=counts, H=None if H is None else H[lo:hi], want_hist=H is not None)
    ev[1].record()
    ev[1].synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def place_decide(ratios, excluded=(), gap=PLACE_GAP, sure=PLACE_SURE):
    """The placement decision from the candidates' slice ratios alone (host logic; tests/test_host_logic.py plays sequences
    through it).  -> (index of the pick or None, verdict):
      "sure"         the lowest ratio is <= `sure`: its store lands in another class than the matrix whatever the others say;
      "two-levels"   the sorted ratios split into two groups at a RELATIVE step >= `gap` (the largest step counts): the groups are
                     two memory classes as this process sees them now -- no absolute level involved -- and the pick is the lowest
                     of the lower group;
      "one-level"    no such step: every candidate lies in one class (the matrix's or another: the probe cannot say) -- the pick
                     is the lowest, and the caller may walk on.
    `excluded`: candidates already found to straddle (confirmation failed) -- never picked, still part of the picture."""
    live = [i for i in range(len(ratios)) if i not in set(excluded)]
    if not live:
        return None, "none"
    order = sorted(live, key=lambda i: ratios[i])
    best = order[0]
    if ratios[best] <= sure:
        return best, "sure"
    allo = sorted(range(len(ratios)), key=lambda i: ratios[i])
    steps = [(ratios[allo[k + 1]] / ratios[allo[k]] - 1.0, k) for k in range(len(allo) - 1)]
    if steps:
        step, k = max(steps)
        if step >= gap:
            lower = [i for i in allo[:k + 1] if i not in set(excluded)]
            if lower:
                return lower[0], "two-levels"
    return best, "one-level"


def placement_enabled():
    """Off with EPILOGOS_PLACEMENT=0, when this torch cannot count a storage's users, and when the launcher put more ranks on the
    node than it has devices (ranks sharing a GPU -- run.py's gloo mode, tests -- must not each hold a home block and a cache of
    probe blocks)."""
    import os
    if os.environ.get("EPILOGOS_PLACEMENT", "1") == "0" or getattr(torch._C, "_storage_Use_Count", None) is None:
        return False
    try:
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        lws = 1
    return lws <= max(torch.cuda.device_count(), 1)


def _order_after_last_user(st):
    """The home is handed out again: whatever the previous holder enqueued on ITS stream must be over before this one's kernels
    write the block.  The views bypass torch's caching allocator, which would do this for an ordinary tensor: an event on the
    previous holder's stream, awaited by the current one (nothing when they are the same stream)."""
    prev, cur = st.get("stream"), torch.cuda.current_stream()
    if prev is not None and prev != cur:
        ev = torch.cuda.Event()
        ev.record(prev)
        cur.wait_event(ev)
    st["stream"] = cur


def alloc_hist(X, N, S):
    """The [R, S] uint16 histogram cache (int16 storage) for a RESIDENT state matrix X -- in another memory CLASS than X once
    the process has shown that it keeps running jobs on that matrix.

    The 288 GB of an MI355X fall into three classes of a third each (96 GB, contiguous in the driver's allocation order -- the
    three ranks of its 12-high HBM stacks would look exactly like this; profiles/r02ae_k1_memory_class_map.txt).  k_bin_hist
    reads X and writes H: with both in one class the launch is 13-17 % slower (2.6 against 2.3 ms for 15 M x 833) whatever the
    offsets, the data or the other buffers are.  HIP does not tell the class of an allocation, so the classifier is the kernel
    itself, on three 1 M-bin slices of X (head, middle, tail): ratio = launches with the H store into the candidate /
    counts-only launches.  Candidate 0 is the PLAIN allocation this function replaces (next to the matrix, as a rule in its
    class); the others are the heads of 4 GiB blocks allocated one after the other and held during the search, which makes the
    driver walk through its memory.  place_decide reads the ratios (a candidate at or under PLACE_SURE ends the walk at once;
    otherwise it goes on until two levels >= 3 % apart show, pick = the lowest of the lower one, or a bound is reached, pick =
    the lowest), and then ONE comparison decides, over the WHOLE matrix (the slices cover a fifth of it and miss a class
    boundary inside it): the pick against the plain allocation -- a pick that does not win by 1 % is set aside and the walk
    goes on; three losses end it.  The cache is therefore never slower than the plain allocation it replaces.

    What the search may cost is tied to what it can earn (0.3 ms per 15 M-bin job):
      * job 1 on a matrix: no search, a plain allocation -- a command-line run (one job per matrix) never pays anything;
      * job PLACE_SEARCH_AT: the QUICK search -- <= 8 blocks, <= 50 ms of probes, ended by the first block the driver is slow to
        hand out: 20-40 ms in all, 160 ms at worst.  About 32 GiB past the matrix come from the driver's pool of cleared memory
        at 0.3 ms per block; the class boundary lies inside that stretch for roughly a third of the fresh processes (it is
        96 GB away at most, anywhere with equal odds), and for another share the plain allocation is in another class already;
      * job PLACE_DEEP_AT, if the plain allocation was kept: the DEEP search, once -- <= 24 blocks (96 GiB, past any class),
        <= 5 s: beyond the cleared pool the driver takes ~30 ms per GiB to hand memory out, i.e. 1-3 s for the walk
        (tools/alloc_probe.py; profiles/r06d_alloc_probe.txt), which a process that keeps running jobs gets back and a single job does not.
    A block that wins is the device's HOME for the life of the process (H is a view of its head): later jobs on the same
    matrix get it without a probe, a job on another matrix after one probe.  The other blocks go back to torch's caching
    allocator (not to the driver: freed device memory is scrubbed in the background at every HBM-bound kernel's expense; torch
    reuses cached blocks and returns them by itself when an allocation would otherwise fail); the report says how much.
    While a view of the home is alive the next request gets a plain allocation; a hand-out on another stream than the previous
    one waits for that stream (_order_after_last_user).  release_placement() gives the home up.
    placement_enabled() says when all of this is off (plain allocations): EPILOGOS_PLACEMENT=0, ranks sharing a GPU.
    EPILOGOS_PLACEMENT_EAGER=1: both searches at the first job (tools, tests)."""
    import os
    R = X.shape[0]
    dev = X.device
    hbytes = R * S * 2
    plain = lambda: torch.empty((R, S), dtype=torch.int16, device=dev)
    if X.numel() < PLACE_MIN_BYTES or not placement_enabled():
        return plain()
    st = _placement.get(dev.index)
    stor = X.untyped_storage()
    key = (stor.data_ptr(), stor.nbytes())
    view = lambda home: home[:hbytes].view(torch.int16).view(R, S)
    eager = os.environ.get("EPILOGOS_PLACEMENT_EAGER") == "1"
    if st is not None and st["home"] is not None and st["home"].numel() >= hbytes:
        users = _storage_users(st["home"])
        if users is None or users > 1:
            st["report"]["plain_while_home_in_use"] = st["report"].get("plain_while_home_in_use", 0) + 1
            return plain()
        if st["key"] == key:
            st["report"]["reuses"] = st["report"].get("reuses", 0) + 1
            _order_after_last_user(st)
            return view(st["home"])
        counts = zeros_counts(S, device=dev)                 # another matrix: is the home good for it too?
        slices = _probe_slices(R)
        r = _probe_ms(X, N, S, st["home"], counts, slices) / _probe_ms(X, N, S, None, counts, slices)
        if r <= PLACE_SURE:
            st["key"] = key
            st["report"].update(revalidated=st["report"].get("revalidated", 0) + 1, ratio=round(r, 3))
            _order_after_last_user(st)
            return view(st["home"])
        st = None                                            # no: this matrix starts over (the old home goes to torch's cache)
        _placement.pop(dev.index, None)
    if st is None or st["key"] != key:
        st = _placement[dev.index] = {"key": key, "home": None, "report": {"jobs_seen": 0, "tier": "none yet"}, "stream": None, "seen": 0, "tier": 0}
    st["seen"] += 1
    st["report"]["jobs_seen"] = st["seen"]
    want = 0
    if st["tier"] < 1 and (eager or st["seen"] >= PLACE_SEARCH_AT):
        want = 1
    elif st["tier"] == 1 and (eager or st["seen"] >= PLACE_DEEP_AT):
        want = 2
    if not want:
        return plain()
    deep = want == 2
    home, h_plain, report = _place_search(X, N, S, int(os.environ.get("EPILOGOS_PLACEMENT_TRIES", PLACE_DEEP_TRIES if deep else PLACE_TRIES)),
                                          3 * PLACE_BUDGET_MS if deep else PLACE_BUDGET_MS, PLACE_DEEP_WALL_MS if deep else 10 * PLACE_BUDGET_MS,
                                          stop_at_slow_alloc=not deep)
    report.update(tier="deep" if deep else "quick", jobs_seen=st["seen"], reuses=0)
    if deep and "quick" not in report:
        report["quick"] = {k: st["report"].get(k) for k in ("decision", "ratios", "blocks_tried", "search_ms")}
    if home is None and report["picked"] == 0 and report["decision"].startswith("plain allocation kept (sure"):
        want = 2                                             # the plain allocation already lies in another class: nothing to look for
    st.update(home=home, report=report, stream=torch.cuda.current_stream(), tier=want)
    if home is None and eager and want == 1:
        return alloc_hist(X, N, S)                           # (eager: the deep search follows at once)
    return view(home) if home is not None else h_plain


PLACE_SLOW_ALLOC_MS = 20.0          # a 4 GiB block that takes longer to get comes from beyond the driver's pool of cleared memory


def _place_search(X, N, S, tries, budget_ms, wall_ms, stop_at_slow_alloc=False):
    """One search of alloc_hist: -> (home block or None = the plain allocation stays, the plain allocation, report).
    stop_at_slow_alloc (the quick search): the walk ends with the first block the driver takes more than PLACE_SLOW_ALLOC_MS to
    hand out -- every further one would cost the same ~120 ms."""
    import time
    R = X.shape[0]
    dev = X.device
    hbytes = R * S * 2
    t_start = time.perf_counter()
    counts = zeros_counts(S, device=dev)
    slices = _probe_slices(R)
    whole = [(0, R)]
    spent = [0.0]                                            # device time of the probes so far (ms)

    def probe(cand, where, reps=2):
        ms = _probe_ms(X, N, S, cand, counts, where, reps=reps)
        spent[0] += ms * (reps + 1)
        return ms

    base = probe(None, slices)
    block = max(PLACE_BLOCK, (hbytes + 4095) // 4096 * 4096)
    h_plain = torch.empty((R, S), dtype=torch.int16, device=dev)
    cands, ratios = [h_plain.view(torch.int8).view(-1)], []
    t0 = probe(cands[0], slices)
    base = min(base, probe(None, slices))                    # (the very first launches of a process run on a cold clock: the lower of two)
    ratios.append(t0 / base)
    whole_ms, excluded, alloc_ms = {}, [], []

    def walk_on():
        """One more block, probed on the slices; False when a bound is reached."""
        k = len(cands) - 1
        if k >= tries or spent[0] >= budget_ms or (time.perf_counter() - t_start) * 1e3 >= wall_ms:
            return False
        if stop_at_slow_alloc and alloc_ms and alloc_ms[-1] > PLACE_SLOW_ALLOC_MS:
            return False
        free, _total = torch.cuda.mem_get_info(dev)
        if free < block + 16 * hbytes + (8 << 30):           # the rest of the job must still fit
            return False
        t0 = time.perf_counter()
        try:
            cands.append(torch.empty(block, dtype=torch.int8, device=dev))
        except RuntimeError:
            return False
        torch.cuda.synchronize(dev)
        alloc_ms.append((time.perf_counter() - t0) * 1e3)
        ratios.append(probe(cands[-1], slices) / base)
        return True

    # walk until the slices decide (or a bound is reached), then the comparison that counts: the pick against the plain allocation
    # over the WHOLE matrix.  A pick that loses is set aside and the walk goes on from where it stopped (three such losses end it).
    pick, verdict = place_decide(ratios, excluded)
    while True:
        while verdict not in ("sure", "two-levels") and walk_on():
            pick, verdict = place_decide(ratios, excluded)
        if pick in (None, 0):
            break
        if slices == whole:
            whole_ms = {0: ratios[0] * base, pick: ratios[pick] * base}
        else:
            if 0 not in whole_ms:
                whole_ms[0] = probe(cands[0], whole, reps=1)
            whole_ms[pick] = probe(cands[pick], whole, reps=1)
        if whole_ms[pick] < whole_ms[0] * (1.0 - PLACE_WIN):
            break
        excluded.append(pick)
        if len(excluded) >= 3 or spent[0] >= 2 * budget_ms:
            pick = 0
            break
        pick, verdict = place_decide(ratios, excluded)
        verdict += " (after %d pick(s) that lost over the whole matrix)" % len(excluded)
    if pick is None or pick in excluded:
        pick = 0
    good = pick != 0
    report = {"probe": "k_bin_hist over 3 x %d bins of the matrix: with the H store into a candidate / counts only; candidate 0 is the plain "
                       "allocation, the others heads of %d GiB blocks allocated one after the other" % (slices[0][1] - slices[0][0], block >> 30),
              "decision": verdict if good else "plain allocation kept (%s)" % verdict, "good": good, "picked": pick, "ratio": round(ratios[pick], 3),
              "ratios": [round(r, 3) for r in ratios], "whole_matrix_ms": {str(i): round(v, 4) for i, v in whole_ms.items()},
              "lost_over_the_whole_matrix": excluded, "blocks_tried": len(cands) - 1,
              "walked_GiB": round(sum(c.numel() for c in cands[1:]) / 2**30, 1),
              "block_alloc_ms": [round(a, 1) for a in alloc_ms],
              "rules": {"sure_at_or_under": PLACE_SURE, "two_levels_apart_by": PLACE_GAP, "must_beat_plain_by": PLACE_WIN,
                        "max_blocks": tries, "max_probe_ms": budget_ms, "max_wall_ms": wall_ms},
              "ms_counts_only": round(base, 4), "probe_device_ms": round(spent[0], 2),
              "left_in_torch_cache_GiB": round(sum(c.numel() for i, c in enumerate(cands) if i and i != pick) / 2**30, 1)}
    home = cands[pick] if good else None
    del cands                                                # losing blocks -> torch's cache (not the driver)
    report["search_ms"] = round((time.perf_counter() - t_start) * 1e3, 2)
    return home, h_plain, report


def placement_report(device=None):
    """What alloc_hist did on this device (None: it never searched)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _placement.get(idx)
    return None if st is None else dict(st["report"])


def release_placement():
    """Forget the home (its memory goes to torch's cache; torch.cuda.empty_cache() then returns it and the probe blocks to the
    driver).  For a process that is done with resident-matrix jobs and wants the memory for something else."""
    _placement.clear()


def states_to_device(x, device="cuda"):
    """Host int array [R, N] of 0-based states -> padded int8 device matrix."""
    x = np.ascontiguousarray(x)
    R, N = x.shape
    ldx = padded_width(N)
    host = np.full((R, ldx), -1, dtype=np.int8)
    host[:, :N] = x.astype(np.int8)
    return torch.from_numpy(host).to(device)


def _check_states(X, N):
    if X.dtype != torch.int8 or X.dim() != 2 or not X.is_cuda or X.stride(1) != 1:
        raise ValueError("state matrix must be a 2-D int8 CUDA tensor with unit column stride")
    if N > X.shape[1]:
        raise ValueError("N exceeds the row width")
    return X.shape[0], X.stride(0)


def zeros_counts(n, dtype=torch.int64, device="cuda"):
    return torch.zeros(n, dtype=dtype, device=device)


def bin_hist(X, N, S, want_hist=True, counts=None, want_counts=True, H=None):
    """K1: per-bin histograms H uint16 [R, S] (as int16 storage) and/or counts[S] += column sums."""
    R, ldx = _check_states(X, N)
    if H is None and want_hist:
        H = torch.empty((R, S), dtype=torch.int16, device=X.device)
    if want_counts and counts is None:
        counts = zeros_counts(S, device=X.device)
    _abi.call("epg_bin_hist", _ptr(X), R, N, ldx, S, _ptr(H), _ptr(counts) if want_counts else None, _stream())
    return H, counts


def bin_hist_s2(X, N, S, counts2=None, H=None, counts=None):
    """K1 with the S2 pair counts of the same bins folded into the launch (epg_bin_hist_s2): -> (H, counts2 int64 [S*S]);
    counts (int64 [S], optional) also gets the state counts."""
    R, ldx = _check_states(X, N)
    if H is None:
        H = torch.empty((R, S), dtype=torch.int16, device=X.device)
    if counts2 is None:
        counts2 = zeros_counts(S * S, device=X.device)
    _abi.call("epg_bin_hist_s2", _ptr(X), R, N, ldx, S, _ptr(H), _ptr(counts), _ptr(counts2), _stream())
    return H, counts2


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None and t.numel() else None for t in ts])


def hist_rows_flat(rows, S, device, dtype=torch.int16):
    """One allocation for the [R_p, S] rows of several parts: -> (flat [R_total_padded, S], starts).  Every part starts 16-byte
    aligned (its first row at a multiple of 8 rows: 8 rows of any S are a multiple of 16 bytes); the <= 7 rows between two parts
    are never written by the count pass (the score pass looks nothing up for them: k_score_s1_from_hist)."""
    starts, at = [], 0
    for r in rows:
        starts.append(at)
        at += (r + 7) // 8 * 8
    return torch.empty((max(at, 1), S), dtype=dtype, device=device), starts


def hist_rows_alloc(rows, S, device):
    """hist_rows_flat as a list of views, one per part."""
    flat, starts = hist_rows_flat(rows, S, device)
    return [flat[a:a + r] for a, r in zip(starts, rows)]


def bin_hist_parts(Xs, Ns, S, counts=None, want_hist=True, Hs=None):
    """K1 over several resident matrices in as few launches as their widths allow (epg_bin_hist_parts; one launch when all
    widths have the same number of 128-byte groups per row): -> (list of H [R_p, S] int16-stored uint16, counts), all parts'
    state counts added into the one `counts` (None: no counts)."""
    n = len(Xs)
    if n == 0:
        return [], counts
    shapes = [_check_states(X, N) if X.shape[0] else (0, max(N, 1)) for X, N in zip(Xs, Ns)]
    dev = Xs[0].device
    if Hs is None and want_hist:
        Hs = hist_rows_alloc([r for r, _l in shapes], S, dev)
    _abi.call("epg_bin_hist_parts", n, _ptr_array(Xs), (C.c_int64 * n)(*[r for r, _l in shapes]), (C.c_int32 * n)(*[int(N or 1) for N in Ns]),
              (C.c_int64 * n)(*[l for _r, l in shapes]), S, _ptr_array(Hs) if Hs is not None else None, _ptr(counts), _stream())
    return Hs, counts


def null_hist_from_binhist_parts(HAs, HBs, n_cols, S, ga, gb, seed, row0s, stream=None):
    """epg_null_hist_from_binhist for several parts in one launch (bit-identical to a call per part); outputs from ONE allocation
    per group.  -> (list of OA, list of OB)."""
    n = len(HAs)
    if n == 0:
        return [], []
    rows = [h.shape[0] for h in HAs]
    dev = HAs[0].device
    OAs, OBs = hist_rows_alloc(rows, S, dev), hist_rows_alloc(rows, S, dev)
    _abi.call("epg_null_hist_from_binhist_parts", n, _ptr_array(HAs), _ptr_array(HBs), (C.c_int64 * n)(*rows), S, n_cols, ga, gb, seed,
              (C.c_int64 * n)(*[int(r) for r in row0s]), _ptr_array(OAs), _ptr_array(OBs),
              _stream() if stream is None else C.c_void_p(stream.cuda_stream))
    return OAs, OBs


def pair_count_null_parts(XAs, XBs, NA, NB, S, seed, row0s, counts=None):
    """Paired mode, default group sizes: count pass of both groups and the null draw of several resident parts in ONE launch
    (epg_pair_count_null_parts).  -> (HAs, HBs, OAs, OBs) lists of [R_p, S] histograms (real groups, null groups).  Raises
    EpilogosHipError(-2) for shapes outside the fused kernel's: bin_hist_parts + null_hist_from_binhist_parts give the same."""
    n = len(XAs)
    sa = [_check_states(X, NA) if X.shape[0] else (0, padded_width(NA)) for X in XAs]
    sb = [_check_states(X, NB) if X.shape[0] else (0, padded_width(NB)) for X in XBs]
    rows = [r for r, _l in sa]
    if rows != [r for r, _l in sb]:
        raise ValueError("paired inputs must have the same number of bins")
    dev = XAs[0].device
    H = hist_rows_alloc(rows + rows, S, dev)
    O = hist_rows_alloc(rows + rows, S, dev)
    _abi.call("epg_pair_count_null_parts", n, _ptr_array(XAs), _ptr_array(XBs), (C.c_int64 * n)(*rows), NA, NB,
              (C.c_int64 * n)(*[l for _r, l in sa]), (C.c_int64 * n)(*[l for _r, l in sb]), S, _ptr_array(H[:n]), _ptr_array(H[n:]),
              _ptr(counts), seed, (C.c_int64 * n)(*[int(r) for r in row0s]), _ptr_array(O[:n]), _ptr_array(O[n:]), _stream())
    return H[:n], H[n:], O[:n], O[n:]


def hist_s2_from_binhist(H, S, counts=None):
    if counts is None:
        counts = zeros_counts(S * S, device=H.device)
    _abi.call("epg_hist_s2_from_binhist", _ptr(H), H.shape[0], S, _ptr(counts), _stream())
    return counts


def hist_s2_from_binhist_pair(HA, HB, S, counts=None):
    """S2 counts of the column concatenation [A|B] from the two groups' histograms (helpers.py:173)."""
    if HA.shape != HB.shape:
        raise ValueError("paired histograms must have the same shape")
    if counts is None:
        counts = zeros_counts(S * S, device=HA.device)
    _abi.call("epg_hist_s2_from_binhist_pair", _ptr(HA), _ptr(HB), HA.shape[0], S, _ptr(counts), _stream())
    return counts


def hist_s3_ws_bytes(R, N, S):
    return max(int(_abi.call("epg_ws_bytes", 3, R, N, S)), 256)


def hist_s3(X, N, S, counts=None, use_workspace=True, ws=None):
    """counts int32 [N*N*S*S] += biosample-pair state co-occurrences.  With a workspace (room for the transposed matrix)
    the matrix-core kernel runs; without one the ABI falls back to the LDS-counter kernel (same integers).  `ws`: a
    caller-owned workspace of at least hist_s3_ws_bytes(R, N, S) bytes (a session reuses one for all its parts)."""
    R, ldx = _check_states(X, N)
    if counts is None:
        counts = zeros_counts(N * N * S * S, dtype=torch.int32, device=X.device)
    if use_workspace:
        if ws is None:
            ws = torch.empty(hist_s3_ws_bytes(R, N, S), dtype=torch.uint8, device=X.device)
        _abi.call("epg_hist_s3", _ptr(X), R, N, ldx, S, _ptr(counts), _ptr(ws), ws.numel(), _stream())
    else:
        _abi.call("epg_hist_s3", _ptr(X), R, N, ldx, S, _ptr(counts), None, 0, _stream())
    return counts


def normalise(counts, q=None, ws=None):
    """q = float32(counts / sum(counts)) on device (expectedCombination.py:42)."""
    if q is None:
        q = torch.empty(counts.numel(), dtype=torch.float32, device=counts.device)
    if ws is None:
        ws = torch.empty(256, dtype=torch.uint8, device=counts.device)
    name = {torch.int64: "epg_normalise_i64", torch.int32: "epg_normalise_i32"}[counts.dtype]
    _abi.call(name, _ptr(counts), counts.numel(), _ptr(q), _ptr(ws), ws.numel(), _stream())
    return q


def _outs(R, S, device, want32, want64, out32=None, out64=None):
    o32 = out32 if out32 is not None else (torch.empty((R, S), dtype=torch.float32, device=device) if want32 else None)
    o64 = out64 if out64 is not None else (torch.empty((R, S), dtype=torch.float64, device=device) if want64 else None)
    return o32, o64


def workspace(saliency, R, N, S, device="cuda"):
    """Caller-owned scratch for the score calls of one saliency (tables; plus a histogram cache when R > 0)."""
    return _ws(saliency, R, N, S, device)


def _ws(saliency, R, N, S, device):
    nbytes = _abi.call("epg_ws_bytes", saliency, R, N, S)
    return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)


def score_s1(X, N, S, q, want32=True, want64=False, out32=None, out64=None, ws=None):
    R, ldx = _check_states(X, N)
    o32, o64 = _outs(R, S, X.device, want32, want64, out32, out64)
    if ws is None:
        ws = _ws(1, R, N, S, X.device)
    _abi.call("epg_score_s1", _ptr(X), R, N, ldx, S, _ptr(q), _ptr(o64), _ptr(o32), _ptr(ws), ws.numel(), _stream())
    return o32, o64


def score_s1_from_binhist(H, N, S, q, want32=True, want64=False, out32=None, out64=None, ws=None):
    R = H.shape[0]
    o32, o64 = _outs(R, S, H.device, want32, want64, out32, out64)
    if ws is None:
        ws = _ws(1, 0, N, S, H.device)
    _abi.call("epg_score_s1_from_binhist", _ptr(H), R, N, S, _ptr(q), _ptr(o64), _ptr(o32), _ptr(ws), ws.numel(),
              _stream())
    return o32, o64


def score_s1_from_binhist_table(H, N, S, T64=None, T32=None, out32=None, out64=None):
    """S1 scores of cached histograms from a caller-built table T[c, s], c = 0..N (device tensors [N + 1, S]; T32 -> float32
    scores, T64 -> float64 scores).  scores.s1ScoreTable builds the pair on the host with the reference's arithmetic."""
    R = H.shape[0]
    for T, dt in ((T64, torch.float64), (T32, torch.float32)):
        if T is not None and (T.dtype != dt or T.numel() != (N + 1) * S or not T.is_contiguous()):
            raise ValueError("S1 table must be a contiguous [N + 1, S] tensor of its output's dtype")
    o32, o64 = _outs(R, S, H.device, T32 is not None, T64 is not None, out32, out64)
    _abi.call("epg_score_s1_from_binhist_table", _ptr(H), R, N, S, _ptr(T64), _ptr(T32), _ptr(o64), _ptr(o32), _stream())
    return o32, o64


def combine_score_s1(counts, H, N, S, q=None, want32=True, want64=False, out32=None, out64=None, ws=None, rezero=False):
    """STEP 2 + STEP 3 of an S1 job in one ABI call: q = normalise(counts) (the all-reduced int64[S] vector), then the
    scores of the cached histograms H.  Returns (q, out32, out64)."""
    R = H.shape[0]
    o32, o64 = _outs(R, S, H.device, want32, want64, out32, out64)
    if q is None:
        q = torch.empty(S, dtype=torch.float32, device=H.device)
    if ws is None:
        ws = _ws(1, 0, N, S, H.device)
    _abi.call("epg_combine_score_s1", _ptr(counts), 1 if rezero else 0, _ptr(H), R, N, S, _ptr(q), _ptr(o64), _ptr(o32),
              _ptr(ws), ws.numel(), _stream())
    return q, o32, o64


def s1_tables(counts, N, S, q=None, ws=None):
    """STEP 2 and the S1 score table of group width N in ONE launch, nothing else (epg_combine_score_s1 with no bins):
    q = normalise(counts) and T[c, s] = kl(c / N, q[s]), c = 0..N.  Returns (q, T64, T32); the tables are views of `ws`."""
    dev = counts.device
    if q is None:
        q = torch.empty(S, dtype=torch.float32, device=dev)
    if ws is None:
        ws = _ws(1, 0, N, S, dev)
    _abi.call("epg_combine_score_s1", _ptr(counts), 0, None, 0, N, S, _ptr(q), None, None, _ptr(ws), ws.numel(), _stream())
    nent = (N + 1) * S
    off = (nent * 8 + 255) // 256 * 256
    return q, ws[:nent * 8].view(torch.float64), ws[off:off + nent * 4].view(torch.float32)


def score_s2(X, N, S, q, perms=None, want32=True, want64=False):
    R, ldx = _check_states(X, N)
    perms = N * (N - 1) if perms is None else perms
    o32, o64 = _outs(R, S, X.device, want32, want64)
    ws = _ws(2, R, N, S, X.device)
    _abi.call("epg_score_s2", _ptr(X), R, N, ldx, S, perms, _ptr(q), _ptr(o64), _ptr(o32), _ptr(ws), ws.numel(),
              _stream())
    return o32, o64


def score_s2_from_binhist(H, N, S, q, perms=None, want32=True, want64=False, out32=None, out64=None, ws=None):
    R = H.shape[0]
    perms = N * (N - 1) if perms is None else perms
    o32, o64 = _outs(R, S, H.device, want32, want64, out32, out64)
    if ws is None:
        ws = _ws(2, 0, N, S, H.device)
    _abi.call("epg_score_s2_from_binhist", _ptr(H), R, N, S, perms, _ptr(q), _ptr(o64), _ptr(o32), _ptr(ws),
              ws.numel(), _stream())
    return o32, o64


def score_s3(X, N, S, q, want32=True, want64=False, ws=None):
    """`ws`: caller-owned workspace of at least epg_ws_bytes(3, R, N, S) bytes (tables + transposed matrix + cells)."""
    R, ldx = _check_states(X, N)
    o32, o64 = _outs(R, S, X.device, want32, want64)
    if ws is None:
        ws = _ws(3, R, N, S, X.device)
    _abi.call("epg_score_s3", _ptr(X), R, N, ldx, S, _ptr(q), _ptr(o64), _ptr(o32), _ptr(ws), ws.numel(), _stream())
    return o32, o64


def pair_scores_s1_from_binhist(HA, HB, HnA, HnB, S, NA, NB, ga, gb, TA, TB, TnA, TnB):
    """Paired S1 in one pass over the four histograms (epg_pair_scores_s1_from_binhist): delta [R, S], null distance [R], STEP 4's
    distance [R] and 1-based largest-difference state [R].  T*: float32 [width + 1, S] device tables (scores.s1ScoreTable).
    Raises EpilogosHipError(-2) when the tables do not fit a CU's LDS; the separate calls give the same results."""
    R = HA.shape[0]
    dev = HA.device
    delta = torch.empty((R, S), dtype=torch.float32, device=dev)
    null = torch.empty(R, dtype=torch.float32, device=dev)
    dist = torch.empty(R, dtype=torch.float32, device=dev)
    maxdiff = torch.empty(R, dtype=torch.int32, device=dev)
    _abi.call("epg_pair_scores_s1_from_binhist", _ptr(HA), _ptr(HB), _ptr(HnA), _ptr(HnB), R, S, NA, NB, ga, gb, _ptr(TA), _ptr(TB), _ptr(TnA),
              _ptr(TnB), _ptr(delta), _ptr(null), _ptr(dist), _ptr(maxdiff), _stream())
    return delta, null, dist, maxdiff


def pair_scores_s1_parts(parts, S, NA, NB, ga, gb, TA, TB, TnA, TnB, qstate=None):
    """epg_pair_scores_s1_parts: `parts` is a list of (HA, HB, HnA, HnB) of the parts' histograms; one launch (per 24 parts) gives
    every part's delta [R, S], null distance [R], STEP 4's distance [R] and largest-difference state [R] and -- with qstate not
    None -- its quiescence mask uint8 [R] (qstate < 0: all zero).  Returns a list of dicts like _HipPairedSession.results_device."""
    n = len(parts)
    if n == 0:
        return []
    dev = parts[0][0].device
    outs = []
    for HA, _HB, _HnA, _HnB in parts:
        R = HA.shape[0]
        outs.append({"delta": torch.empty((R, S), dtype=torch.float32, device=dev), "null": torch.empty(R, dtype=torch.float32, device=dev),
                     "rdist": torch.empty(R, dtype=torch.float32, device=dev), "mdiff": torch.empty(R, dtype=torch.int32, device=dev),
                     "quies": torch.empty(R, dtype=torch.uint8, device=dev) if qstate is not None else None})
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() if t is not None and t.numel() else None for t in ts])
    rows = (C.c_int64 * n)(*[p[0].shape[0] for p in parts])
    _abi.call("epg_pair_scores_s1_parts", n, arr([p[0] for p in parts]), arr([p[1] for p in parts]), arr([p[2] for p in parts]),
              arr([p[3] for p in parts]), rows, S, NA, NB, ga, gb, _ptr(TA), _ptr(TB), _ptr(TnA), _ptr(TnB), arr([o["delta"] for o in outs]),
              arr([o["null"] for o in outs]), arr([o["rdist"] for o in outs]), arr([o["mdiff"] for o in outs]),
              arr([o["quies"] for o in outs]) if qstate is not None else None, -1 if qstate is None else int(qstate), _stream())
    return outs


def pair_finish(a, b, want_dist=True):
    R, S = a.shape
    delta = torch.empty_like(a)
    dist = torch.empty(R, dtype=torch.float32, device=a.device) if want_dist else None
    _abi.call("epg_pair_finish", _ptr(a), _ptr(b), R, S, _ptr(delta), _ptr(dist), _stream())
    return delta, dist


def pair_metrics(delta, roundtrip=True):
    """Signed squared distance float32[R] and 1-based largest-|delta| state int32[R] (roiAndVisualPairwise.py:347-354)."""
    R, S = delta.shape
    dist = torch.empty(R, dtype=torch.float32, device=delta.device)
    maxdiff = torch.empty(R, dtype=torch.int32, device=delta.device)
    _abi.call("epg_pair_metrics", _ptr(delta), R, S, 1 if roundtrip else 0, _ptr(dist), _ptr(maxdiff), _stream())
    return dist, maxdiff


def quiescent(XA, NA, XB, NB, qstate):
    R, ldxa = _check_states(XA, NA)
    R2, ldxb = _check_states(XB, NB)
    if R != R2:
        raise ValueError("paired inputs must have the same number of bins")
    mask = torch.empty(R, dtype=torch.uint8, device=XA.device)
    _abi.call("epg_quiescent", _ptr(XA), NA, ldxa, _ptr(XB), NB, ldxb, R, qstate, _ptr(mask), _stream())
    return mask


def null_hist(XA, NA, XB, NB, S, ga, gb, seed, row0=0):
    R, ldxa = _check_states(XA, NA)
    _, ldxb = _check_states(XB, NB)
    HA = torch.empty((R, S), dtype=torch.int16, device=XA.device)
    HB = torch.empty((R, S), dtype=torch.int16, device=XA.device)
    _abi.call("epg_null_hist", _ptr(XA), NA, ldxa, _ptr(XB), NB, ldxb, R, S, ga, gb, seed, row0, _ptr(HA), _ptr(HB),
              _stream())
    return HA, HB


_hip_rt = None


def _pinned_bytes(nbytes):
    """A page-locked int8 host tensor of EXACTLY nbytes, from hipHostMalloc through ctypes.  torch.empty(pin_memory=True) rounds
    every request up to a power of two (its caching host allocator): the 1.11 GB staging buffers of a whole-genome run became
    2 GiB each -- twice the memory to lock (0.15-0.6 s per buffer, during which every thread of the process that faults a page
    stands still) and to give back at exit.  torch recognises the memory as pinned (it asks the driver), so copies from it are
    asynchronous.  The buffers live as long as the process."""
    global _hip_rt
    if _hip_rt is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip_rt = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip_rt is None:
            _hip_rt = False
    if _hip_rt:
        import weakref
        ptr = C.c_void_p()
        if _hip_rt.hipHostMalloc(C.byref(ptr), C.c_size_t(int(nbytes)), C.c_uint(0)) == 0 and ptr.value:
            # numpy array -> ctypes array (its base) -> the page-locked memory; the memory goes back to the driver when the last
            # of them -- i.e. the last tensor or view made from it -- is gone (a process that runs several jobs must not keep
            # 9 GB of page-locked memory per job)
            base = (C.c_int8 * int(nbytes)).from_address(ptr.value)
            weakref.finalize(base, _hip_rt.hipHostFree, C.c_void_p(ptr.value))
            t = torch.from_numpy(np.frombuffer(base, dtype=np.int8))
            if t.is_pinned():
                return t
            del t                                        # (the finalizer frees it)
    return torch.empty(int(nbytes), dtype=torch.int8, pin_memory=True)


class PinnedPool:
    """A few page-locked host staging buffers.  in_order=True hands them out IN TICKET ORDER (ticket k waits until tickets
    < k were served and a buffer is free): for a consumer that takes the parts in ticket order, parser threads running
    ahead can neither exhaust the pool nor starve the part it waits for.  in_order=False serves whoever asks first -- for a
    consumer that takes the parts as they complete (driver._stream_parts).  Buffers grow to the largest request and are
    reused for the whole run."""

    def __init__(self, n_buffers=3, in_order=True):
        import threading
        self.cv = threading.Condition()
        self.free = [None] * n_buffers          # None = not allocated yet
        self.next = 0
        self.in_order = in_order
        self.aborted = False
        self.weights = {}
        self.target = 0
        self.waiting = set()
        self.closed = False

    def hint(self, weights):
        """{ticket: bytes of the input file behind it}.  Page-locking memory is slow (~3 GB/s): a pool whose buffers grow to each
        larger request re-allocated ~1 GB a dozen times while the files of a genome completed smallest first (profiles/r04h:
        1.5 s between the end of chr1's parse and its hand-over).  With the hint, the first request scales its own size by
        (largest file / its file) and every buffer is allocated ONCE, at that size (+ 5 %)."""
        self.weights = dict(weights)

    def abort(self):
        """The consumer gave up (an error): wake every waiting parser thread instead of leaving it blocked."""
        with self.cv:
            self.aborted = True
            self.cv.notify_all()

    def acquire(self, ticket, nbytes):
        import os, sys, time
        trace = os.environ.get("EPILOGOS_TIMING") == "2"
        t_in = time.perf_counter()
        with self.cv:
            # among the readers waiting for a buffer the one with the LARGEST file goes first: the largest files finish last
            # and nothing can start behind them -- they are the critical path of the parse stage
            self.waiting.add(ticket)
            first = lambda: not self.weights or self.weights.get(ticket, 0) >= max(self.weights.get(t, 0) for t in self.waiting)
            self.cv.wait_for(lambda: self.aborted or ((not self.in_order or self.next == ticket) and len(self.free) > 0 and first()))
            self.waiting.discard(ticket)
            if self.aborted:
                raise RuntimeError("staging pool aborted")
            buf = self.free.pop()
            self.next += 1
            self.cv.notify_all()
        w = self.weights.get(ticket)
        if w and nbytes:
            self.target = max(self.target, int(nbytes / w * max(self.weights.values()) * 1.05))
        t_got = time.perf_counter()
        grew = 0
        if buf is None or buf.numel() < nbytes:
            buf = None                                   # (give the smaller buffer back before asking for the larger one)
            buf = _pinned_bytes(max(int(nbytes), self.target, 1 << 20))
            grew = buf.numel()
        if trace:
            print("    [pool] ticket %2d: waited %.2f s for a staging buffer%s  (at %.3f)" % (
                ticket, t_got - t_in, ", page-locked %.2f GB in %.2f s" % (grew / 1e9, time.perf_counter() - t_got) if grew else "",
                time.perf_counter()), file=sys.stderr, flush=True)
        return buf

    def skip(self, ticket):
        """A part that failed before asking for its buffer must not block the tickets behind it."""
        if not self.in_order:
            return
        with self.cv:
            self.cv.wait_for(lambda: self.aborted or self.next >= ticket)
            if self.next == ticket:
                self.next += 1
            self.cv.notify_all()

    def release(self, buf):
        with self.cv:
            if self.closed:
                return                                   # (dropped: the memory goes back to the driver with this last reference)
            self.free.append(buf)
            self.cv.notify_all()

    def close(self, background=True):
        """Nothing more will be staged: give the page-locked buffers back now -- on a helper thread, so that un-locking ~9 GB
        passes under the score pass and the writers instead of at process exit (0.3-0.4 s of a whole-genome run).  Buffers still
        out with an upload in flight are dropped when they come back."""
        import threading
        with self.cv:
            self.closed = True
            bufs, self.free = [b for b in self.free if b is not None], []
        if not bufs:
            return

        def work(bufs=bufs):
            while bufs:
                bufs.pop()
        if background:
            threading.Thread(target=work, name="epilogos-staging-close", daemon=True).start()
        else:
            work()


def upload_states(pinned, R, ldx, copy_stream, device="cuda"):
    """Asynchronous H2D copy of a staged [R, ldx] int8 matrix on `copy_stream`.  Returns (X, event): the current stream
    must wait for the event before reading X; the staging buffer may be reused once the event has completed.  X is allocated
    on the copy stream (the caching allocator then knows the copy may start at once) and recorded for the current stream,
    which is where the kernels that read it run."""
    with torch.cuda.stream(copy_stream):
        X = torch.empty((R, ldx), dtype=torch.int8, device=device)
        if R:
            X.copy_(pinned[:R * ldx].view(R, ldx), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(copy_stream)
    X.record_stream(torch.cuda.current_stream())
    return X, ev


def quiescent_from_binhist(HA, NA, HB, NB, S, qstate):
    """Quiescence mask from the two groups' histograms (scores.py:294-303)."""
    R = HA.shape[0]
    mask = torch.empty(R, dtype=torch.uint8, device=HA.device)
    _abi.call("epg_quiescent_from_binhist", _ptr(HA), _ptr(HB), R, S, NA, NB, qstate, _ptr(mask), _stream())
    return mask


def null_hist_from_binhist(HA, HB, n_cols, S, ga, gb, seed, row0=0, stream=None):
    """Histograms of the two shuffled null groups from the REAL groups' histograms (multivariate hypergeometric, exact).
    stream: a torch.cuda.Stream to launch on instead of the current one (the outputs are allocated from the CURRENT stream's
    pool all the same; the caller orders the streams with events)."""
    if HA.shape != HB.shape:
        raise ValueError("paired histograms must have the same shape")
    OA, OB = torch.empty_like(HA), torch.empty_like(HB)
    _abi.call("epg_null_hist_from_binhist", _ptr(HA), _ptr(HB), HA.shape[0], S, n_cols, ga, gb, seed, row0, _ptr(OA), _ptr(OB),
              _stream() if stream is None else C.c_void_p(stream.cuda_stream))
    return OA, OB


def hist_to_numpy(H):
    """int16-stored uint16 histogram tensor -> numpy uint16."""
    return H.cpu().numpy().view(np.uint16)

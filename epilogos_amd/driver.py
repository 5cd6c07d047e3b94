"""Genome-wide driver: replaces the reference's per-chromosome SLURM fan-out (run.py:190-279) with a contiguous
bin-range partition across the GPUs of one node and a single all-reduce of the state-count array.

The bins of all input files, concatenated in file order, are split with the reference's own splitRows rule
(helpers.py:116-118): rank g of G owns global bins [g*R//G, (g+1)*R//G).  Pass 1 counts locally, one
all-reduce(SUM) over torch.distributed (backend nccl = RCCL over xGMI on GPUs; gloo in the CPU tests) makes the
counts global, every rank normalises identically, pass 2 scores locally with no further communication.  Integer
sums make the result independent of G and of the reduction order.

Data flow of a rank (the pipeline bench.py times, plus the host ends):
  parser threads (files in parallel, native inflate + parse straight into pinned, row-padded staging buffers)
  -> ONE asynchronous H2D copy per part -> count pass (k_bin_hist ...) whose per-bin histograms (S1/S2) or state matrix
  (S3, paired) STAY in HBM -> all-reduce -> count check + normalise -> score pass from the resident data -> D2H of the
  float32 scores -> native "%.5f" + gzip writer threads.  Host memory holds the parts in flight, not the genome.
Outputs keep the reference's names.  A file that lies inside one rank's range is written by that rank under its final
name; a file cut by a range border is written as gzip members named from the partition plan (file index + first row)
and concatenated by rank 0 (a multi-member gzip file is a valid gzip file).  The arrays STEP 4 needs travel to rank 0
through torch.distributed send/recv, not through the file system."""
import os
import sys
import time
import shutil
from concurrent.futures import ThreadPoolExecutor
from contextlib import closing
from pathlib import Path

import numpy as np

from . import _io
from . import backend as _backend
from .helpers import fileStem, readStates, readTable, splitRows
from .scores import writeScores


def shuffle_key(file_index, row):
    """Philox counter base of a bin in the paired-mode null shuffle: (file ordinal, row in the file) packed as
    file << 40 | row.  The reference's shuffle is unseeded (helpers.py:183), so any key that does not depend on the partition
    will do; this one is known the moment a file has been parsed -- round 3 keyed by the GLOBAL bin index, which the one-rank
    command line only learns after the last file -- and the outputs stay identical for 1, 2, ... N GPUs."""
    return (int(file_index) << 40) + int(row)


def plan_partition(rows_per_file, world):
    """[(file_index, lo, hi)] per rank: rank ranges from splitRows on the concatenated bins, cut at file borders."""
    total = int(sum(rows_per_file))
    starts = np.concatenate([[0], np.cumsum(rows_per_file)]).astype(np.int64)
    plans = []
    for (g_lo, g_hi) in splitRows(total, world):
        parts = []
        for f, n in enumerate(rows_per_file):
            lo, hi = max(g_lo, starts[f]), min(g_hi, starts[f + 1])
            if lo < hi:
                parts.append((f, int(lo - starts[f]), int(hi - starts[f])))
        plans.append(parts)
    return plans


class _Timer:
    """EPILOGOS_TIMING=1 prints the wall time of the driver's phases on rank 0."""

    def __init__(self, rank):
        import time
        self.on = rank == 0 and bool(os.environ.get("EPILOGOS_TIMING"))
        self.time = time.perf_counter
        self.t = self.time()

    def lap(self, label):
        if self.on:
            now = self.time()
            print("    [timing] %-34s %7.2f s" % (label, now - self.t), flush=True)
            self.t = now

    def note(self, text):
        if self.on:
            print("    [timing] " + text, flush=True)


_DTYPES = [np.float32, np.uint8, np.int64, np.int32, np.bool_, np.int16, np.int8]


class _Dist:
    """Thin wrapper so that the single-process case needs no process group."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self.comm_device = None
        if self.dist and self.dist.get_backend() == "nccl":         # RCCL moves device tensors only
            import torch
            self.comm_device = torch.device("cuda", torch.cuda.current_device())

    def all_reduce_counts(self, counts, device=None):
        """SUM-all-reduce an integer numpy array (host-array backends; the product backend reduces its device tensor)."""
        if not self.dist:
            return counts
        import torch
        t = torch.from_numpy(np.ascontiguousarray(counts))
        dev = device if device is not None else self.comm_device
        if dev is not None:
            t = t.to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().reshape(counts.shape)

    def all_reduce_tensor(self, t):
        """SUM-all-reduce a tensor in place where it lives (device tensors go over RCCL/xGMI)."""
        if self.dist:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_ints(self, values):
        """Element-wise maximum of a short list of non-negative integers over the ranks (a rank without files learns the
        column counts this way instead of opening a file)."""
        if not self.dist:
            return [int(v) for v in values]
        import torch
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        if self.comm_device is not None:
            t = t.to(self.comm_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [int(v) for v in t.cpu().tolist()]

    # point-to-point hand-over of arrays / tensors: a small header (dtype, shape), then the bytes.  Device tensors travel
    # device to device over RCCL; with a host-side backend (gloo) they are staged through the host.
    def send_tensors(self, tensors, dst):
        import torch
        for t in tensors:
            if isinstance(t, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(t))
            if t.dim() > 2:
                raise ValueError("send_tensors: at most two dimensions")
            t = t.contiguous()
            np_dtype = np.dtype(str(t.dtype).replace("torch.", "")).type
            shape = list(t.shape) + [1] * (2 - t.dim())
            head = torch.tensor([_DTYPES.index(np_dtype), t.dim()] + shape, dtype=torch.int64)
            self.dist.send(head.to(self.comm_device) if self.comm_device is not None else head, dst)
            if t.numel():
                flat = t.reshape(-1).view(torch.uint8)
                flat = flat.to(self.comm_device) if self.comm_device is not None else flat.cpu()
                self.dist.send(flat, dst)

    def recv_tensors(self, n, src):
        """-> n tensors, on the communicator's device (RCCL) or on the host (gloo)."""
        import torch
        out = []
        for _ in range(n):
            head = torch.zeros(4, dtype=torch.int64, device=self.comm_device)
            self.dist.recv(head, src)
            code, ndim, d0, d1 = (int(v) for v in head.cpu().tolist())
            dtype = getattr(torch, np.dtype(_DTYPES[code]).name)
            shape = (d0, d1)[:ndim]
            nbytes = int(np.prod(shape)) * np.dtype(_DTYPES[code]).itemsize
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.comm_device)
            if nbytes:
                self.dist.recv(buf, src)
            out.append(buf.view(dtype).reshape(shape))
        return out

    def send_arrays(self, arrays, dst):
        self.send_tensors([np.ascontiguousarray(a) for a in arrays], dst)

    def recv_arrays(self, n, src):
        return [t.cpu().numpy() for t in self.recv_tensors(n, src)]


def _open_single(be, S, saliency):
    return be.open_single(S, saliency)


def _open_paired(be, S, saliency, quiescentState, groupSize, seed):
    return be.open_paired(S, saliency, quiescentState, groupSize, seed)


# ---- input side
def _check_range(path, rng, numStates):
    """The reference indexes a numStates-long array with (file value - 1) and dies on anything outside the model
    (expected.py:113 IndexError); a state model smaller than the data must not run to completion here either."""
    lo, hi = rng
    if (lo, hi) != (0, 0) and (lo < 1 or hi > numStates):
        raise ValueError("{}: state values {}..{} are outside the {}-state model (1..{})".format(path, lo, hi, numStates, numStates))


class _Readers:
    """The parser threads of a rank.  jobs: [(path, lo, hi or None)].  Files are inflated and parsed in threads (inflate is
    serial per file, files run in parallel; the native reader releases the GIL); `parts()` yields (ticket = index in jobs, states
    [rows, width], N, Locations) AS THE PARTS COMPLETE.  Destinations come from the session (page-locked staging, the reader with
    the largest file first) -- which need not exist yet when the threads start: a reader asks for its destination only after
    its file's inflate and line count, so the command line starts them BEFORE it imports torch and initialises the GPU (about a
    second of a whole-genome run) and attaches the session later (`attach`).
    Scheduling: one worker per usable core, largest files first.  Round 2 measured the whole genome on 16 cores with 24
    workers and parts handed over in file order: every part waited behind chr1, whose inflate -- the longest job, started
    together with 23 others on 16 cores -- took 7 s instead of 3; the count pass does not care about the order."""

    def __init__(self, jobs):
        import threading
        self.jobs = [(Path(p), lo, hi) for p, lo, hi in jobs]
        self.sess = None
        self.ready = threading.Event()
        self.failed = False
        self.trace = os.environ.get("EPILOGOS_TIMING") == "2"        # per-part timeline on stderr
        self.t_origin = time.perf_counter()
        # This rank's share of the host (eight ranks that each sized their pools from the whole node ran 120
        # native threads on a 16-core quota -- the cgroup then throttles every process for the rest of each 100 ms period).
        ncores = _io.host_budget()
        self.workers = max(1, min(int(os.environ.get("EPILOGOS_PARSE_WORKERS", ncores)), max(len(self.jobs), 1)))
        self.left = len(self.jobs)
        self.pool = ThreadPoolExecutor(max_workers=self.workers)
        _io.set_reader_plan(min(self.workers, self.left))      # files being read side by side: what the readers share the cores by
        order = sorted(range(len(self.jobs)), key=lambda t: (-self._weight(t), t))
        self.futs = [self.pool.submit(self._read, t) for t in order]

    def _weight(self, ticket):                         # bytes of input behind a job (a row range: unknown share, the whole file)
        try:
            return os.path.getsize(self.jobs[ticket][0])
        except OSError:
            return 0

    def attach(self, sess):
        """The session whose staging buffers the readers parse into (or any object with alloc(ticket) / skip(ticket))."""
        if hasattr(sess, "pool") and all(hi is None for _p, _lo, hi in self.jobs):
            # whole text files: their sizes tell the staging pool how large its buffers will have to become (PinnedPool.hint)
            sess.pool.hint({t: max(self._weight(t), 1) for t in range(len(self.jobs))})
        self.sess = sess
        self.ready.set()

    def abort(self):
        """Give up: wake every reader that waits for a session or a staging buffer, drop the jobs that have not started."""
        self.failed = True
        self.ready.set()
        if self.sess is not None and hasattr(self.sess, "pool"):
            self.sess.pool.abort()
        for f in self.futs:
            f.cancel()

    def _read(self, ticket):
        path, lo, hi = self.jobs[ticket]
        N = [None]
        t_begin = time.perf_counter() - self.t_origin

        def alloc(R, n):                               # called by the native reader once it knows the file's shape
            N[0] = n
            self.ready.wait()
            if self.failed:
                raise RuntimeError("readers aborted")
            a0 = self.sess.alloc(ticket)
            return a0(R, n) if a0 is not None else np.empty((R, n), dtype=np.int8)
        try:
            # native threads per file: 0 = "share" -- every parallel phase of a file takes this rank's budget divided by the
            # files being read at that moment (one thread each while sixteen files inflate side by side, more for the last,
            # largest files).  (With every reader fanning its short parse phases out to all cores next to fifteen inflating
            # threads, a cgroup CPU quota throttles the whole process for the rest of each 100 ms period: phases that take
            # 0.03 s alone took 1 s.)
            arr, loc, rng = readTable(path, None if hi is None else (lo, hi), alloc=alloc, with_range=True, threads=0)
        except BaseException:
            if self.sess is not None:
                self.sess.skip(ticket)
            raise
        if self.trace:                                     # (absolute CLOCK_MONOTONIC seconds too: they line up with [epgio])
            print("    [part %2d] %-28s reader %6.2f .. %6.2f s  (%.3f .. %.3f)" % (ticket, Path(path).name[:28], t_begin,
                  time.perf_counter() - self.t_origin, t_begin + self.t_origin, time.perf_counter()), file=sys.stderr, flush=True)
        return ticket, arr, N[0], loc, rng

    def parts(self, numStates):
        from concurrent.futures import as_completed
        try:
            for f in as_completed(self.futs):
                self.left -= 1
                _io.set_reader_plan(min(self.workers, self.left))
                t, arr, N, loc, rng = f.result()
                _check_range(self.jobs[t][0], rng, numStates)
                t_yield = time.perf_counter() - self.t_origin
                yield t, arr, N, loc
                if self.trace:
                    print("    [part %2d] consumed %6.2f .. %6.2f s" % (t, t_yield, time.perf_counter() - self.t_origin), file=sys.stderr, flush=True)
        except BaseException:
            self.abort()
            raise
        finally:
            _io.set_reader_plan(0)
            self.pool.shutdown(wait=True)


_early = None


def start_readers_early(jobs):
    """Called by the command line before it imports torch: the rank's readers start on `jobs` now; the stage driver picks them
    up if it arrives at the same job list (`_stream_parts`), else they are aborted and fresh ones start."""
    global _early
    _early = _Readers(jobs)
    return _early


def abort_early_readers():
    global _early
    if _early is not None:
        _early.abort()
        _early = None


def _stream_parts(jobs, sess, numStates):
    """-> generator of (ticket, states, N, Locations) in order of completion (see _Readers)."""
    global _early
    jobs = [(Path(p), lo, hi) for p, lo, hi in jobs]
    readers, _early = _early, None
    if readers is not None and readers.jobs != jobs:
        readers.abort()
        readers.pool.shutdown(wait=True)
        readers = None
    if readers is None:
        readers = _Readers(jobs)
    readers.attach(sess)
    return readers.parts(numStates)


def _columns_of(path):
    return readStates(file1Path=path, rowsToCalc=(0, 1), verbose=False).shape[1]


def _part_name(kind, fileTag, fi, lo):
    return ".part_{}_{}_f{:04d}_{:012d}.gz".format(kind, fileTag, fi, lo)


def _clean_parts(outputDir, kind, fileTag):
    """Part files of a crashed earlier run with the same tag must not be mistaken for this run's."""
    import re
    pat = re.compile(r"^\.part_{}_{}_f\d{{4}}_\d{{12}}\.gz$".format(re.escape(kind), re.escape(fileTag)))
    for p in outputDir.iterdir():
        if pat.match(p.name):
            p.unlink()


def _assemble_text(outputDir, kind, final_name, fileTag, fi, owners, rows_fi):
    """Rank 0: a file cut by range borders is the concatenation of its parts' gzip members, in row order, by the exact
    names the plan gives; a file with one owner was written under its final name already."""
    if len(owners) == 1 and owners[0] == (0, rows_fi):
        return
    with open(outputDir / final_name, "wb") as out:
        for lo, _hi in owners:
            p = outputDir / _part_name(kind, fileTag, fi, lo)
            with open(p, "rb") as src:
                shutil.copyfileobj(src, out)
            p.unlink()


def _cat(arrs, empty):
    if len(arrs) == 1:
        return arrs[0]                                 # (a file that lies in one rank's range: no copy of its 70 B per bin)
    return np.concatenate(arrs) if arrs else empty


def _cat_locations(locs):
    if len(locs) == 1:
        return locs[0]
    blobs = [np.asarray(l.blob) for l in locs]
    offs, base = [np.zeros(1, dtype=np.int64)], 0
    for l in locs:
        offs.append(np.asarray(l.offsets[1:]) + base)
        base += int(l.offsets[-1])
    return _io.Locations(_cat(blobs, np.zeros(0, dtype=np.uint8)), np.concatenate(offs))


def _gather_parts(d, plans, my_payloads, n_arrays):
    """Rank 0 collects every part's arrays in plan order: {(file, lo): [arrays]}; the other ranks send theirs."""
    if d.rank != 0:
        for payload in my_payloads:
            d.send_arrays(payload, 0)
        return None
    got = {}
    for r, parts in enumerate(plans):
        for k, (fi, lo, hi) in enumerate(parts):
            got[(fi, lo)] = my_payloads[k] if r == 0 else d.recv_arrays(n_arrays, r)
    return got


def _cached_rows(path):
    """Rows of an input file if the --cache-dir side-car of an earlier run knows them (no pass over the file), else None."""
    from .helpers import _cache_paths
    cache = _cache_paths(path)
    if cache is None or not all(c.exists() for c in cache):
        return None
    return int(np.load(cache[2], mmap_mode="r").shape[0]) - 1


def _assign_files(files, world, sizes=None):
    """Parser rank of every file while the row counts are still unknown.  Inflating and parsing a file is one serial stream
    (plus the count pass over it, which for S3 is the bulk of the job), so what has to be even is the BYTES per rank: longest
    file first, each to the rank with the least so far (LPT) -- hg19's 24 files over 8 ranks: the fullest rank holds 1.04 x the
    mean.  (Round 3 gave every file to the rank whose bin range holds most of it, so that only border pieces changed hands:
    1.40 x the mean at 8 ranks; what changes hands is 36 B per bin for S1 / S2 / paired and one state row for S3 -- device to
    device, a fraction of a second for a genome -- while an unbalanced parse costs seconds.)  Ties go to the rank whose
    estimated bin range holds most of the file, which keeps the small cases -- a file or two per rank -- local."""
    if sizes is None:
        sizes = [max(os.path.getsize(f), 1) for f in files]
    local, share = [0] * len(files), [-1] * len(files)
    for g, parts in enumerate(plan_partition(sizes, world)):
        for fi, lo, hi in parts:
            if hi - lo > share[fi]:
                share[fi], local[fi] = hi - lo, g
    load, owner = [0] * world, [0] * len(files)
    for fi in sorted(range(len(files)), key=lambda k: (-sizes[k], k)):
        least = min(load)
        g = local[fi] if load[local[fi]] == least else load.index(least)
        owner[fi] = g
        load[g] += sizes[fi]
    return owner


def _plan(files, d, tm, files2=None):
    """-> (mode, rows per file or None, jobs [(file index, lo, hi or None)] of this rank, parser rank per file or None).
    "whole":    one rank; whole files, the row counts come out of the parse.
    "ranges":   several ranks and every row count is known from the --cache-dir side-cars: each rank reads exactly its own
                row ranges from the cache; nothing is inflated, nothing changes hands.
    "assigned": several ranks, text inputs.  Every file is inflated and parsed ONCE, by ONE rank (_assign_files), which
                also runs the count pass over it -- the counts are summed over the ranks anyway, whoever owns the bins.
                The row counts are then exchanged (one all-reduce of a vector with an entry per file), the exact bin-range
                partition follows, and what the score pass needs of the rows a rank parsed for another (per-bin histograms
                for S1 / S2 / paired, state rows for S3) is handed over device to device.  (Round 2 let EVERY rank gunzip
                EVERY file just to count its lines before the first byte was parsed: the reference's helpers.py:154
                re-read, one level up.)"""
    F = len(files)
    if d.world == 1:
        return "whole", None, [(fi, 0, None) for fi in range(F)], None
    rows = [_cached_rows(f) for f in files]
    cached = all(r is not None for r in rows) and all(_cached_rows(f) is not None for f in (files2 or []))
    # The two modes run different collective sequences, so the ranks must agree: "ranges" only if EVERY rank sees every
    # side-car (a cache being filled or cleaned by another run, per-rank file-system views) and the same row counts.
    mine = [int(r) for r in rows] if cached else [0] * F
    agree = d.max_ints([0 if cached else 1] + mine + [-r for r in mine])     # max and (negated) min of every row count
    if agree[0] == 0 and agree[1:1 + F] == mine and agree[1 + F:] == [-r for r in mine]:
        return "ranges", rows, plan_partition(rows, d.world)[d.rank], None
    owner = _assign_files(files, d.world)
    return "assigned", None, [(fi, 0, None) for fi in range(F) if owner[fi] == d.rank], owner


def _exchange_rows(d, jobs, locs, F):
    """Row count of every file from the ranks that parsed them: one all-reduce of an int64 vector."""
    vec = np.zeros(F, dtype=np.int64)
    for t, (fi, _lo, _hi) in enumerate(jobs):
        vec[fi] = len(locs[t])
    return [int(v) for v in d.all_reduce_counts(vec)]


def _redistribute(d, sess, plans, owner, mine, starts, widths):
    """"assigned" mode, after the exact plan is known.  mine: {file index: (session part id of the whole file, Locations)}
    of the files this rank parsed.  Walks every part of every rank's range in ONE global order (all ranks run the same
    loop, so each blocking send meets its receive): a part whose parser is its owner becomes a row slice of the resident
    data; otherwise the parser sends what the score pass needs plus the rows' coordinates, and the owner takes them in.
    Returns the session part ids and the Locations of this rank's parts, in plan order."""
    pid_of, loc_of = {}, {}
    for g, parts in enumerate(plans):
        for fi, lo, hi in parts:
            p = owner[fi]
            row0 = shuffle_key(fi, lo)
            if p == g:
                if d.rank == g:
                    pid, loc = mine[fi]
                    part = loc.slice(lo, hi)
                    pid_of[(fi, lo)] = sess.slice_part(pid, lo, hi, row0)
                    loc_of[(fi, lo)] = _io.Locations(np.ascontiguousarray(part.blob), np.ascontiguousarray(part.offsets))
            elif d.rank == p:
                pid, loc = mine[fi]
                part = loc.slice(lo, hi)
                d.send_tensors(list(sess.export_rows(pid, lo, hi)) + [np.ascontiguousarray(part.blob), np.ascontiguousarray(part.offsets)], g)
            elif d.rank == g:
                got = d.recv_tensors(sess.n_export + 2, p)
                pid_of[(fi, lo)] = sess.import_rows(got[:-2], widths, row0)
                loc_of[(fi, lo)] = _io.Locations(got[-2].cpu().numpy(), got[-1].cpu().numpy())
    for pid, _loc in mine.values():                    # the whole-file entries; the slices keep alive what they use
        sess.drop_part(pid)
    my = plans[d.rank]
    return [pid_of[(fi, lo)] for fi, lo, _hi in my], [loc_of[(fi, lo)] for fi, lo, _hi in my]


_deferred = []


def _wait_writes(writer, jobs):
    try:
        for j in jobs:
            j.result()
    finally:
        writer.shutdown(wait=True)


def finish_writes():
    """Wait for the text writers a run with defer_writes=True left running (and re-raise what one of them raised)."""
    while _deferred:
        writer, jobs, tm, label = _deferred.pop()
        _wait_writes(writer, jobs)
        tm.lap(label)


def run_single_group(files, numStates, saliency, outputDir, fileTag, verbose=False, backend=None, device=None,
                     keep_temp_scores=True, defer_writes=False):
    """STEP 1-3 for a single group over `files` (one per chromosome).  Returns (exp_freq float32, results) where results
    (rank 0 only, else None) maps file stem -> (chrName, float32 scores [R, S], _io.Locations) for an in-process STEP 4.
    keep_temp_scores writes the reference's temp_scores_{tag}_{stem}.npz (scores.py:166-169) for a STEP 4 run
    elsewhere; the command line skips them because its STEP 4 would delete them a moment later."""
    be = backend if backend is not None else _backend.get()
    _io.set_state_limit(numStates)                     # above 31 states the parser keeps values up to 127 (wide kernels)
    d = _Dist()
    files = [Path(f) for f in files]
    outputDir = Path(outputDir)
    tm = _Timer(d.rank)
    if d.rank == 0:
        _clean_parts(outputDir, "scores", fileTag)
    d.barrier()
    mode, rows, my_parts, owner = _plan(files, d, tm)
    sess = _open_single(be, numStates, saliency)

    # STEP 1: every part of this rank is parsed, uploaded once and counted; what the score pass needs stays resident
    pids, locs, N = [None] * len(my_parts), [None] * len(my_parts), None
    with closing(_stream_parts([(files[fi], lo, hi) for fi, lo, hi in my_parts], sess, numStates)) as stream:
        for t, arr, n, loc in stream:                  # in order of completion; t = index in my_parts
            N = max(N or 0, n)                         # an empty file has no width: it must not be the one that is remembered
            pids[t] = sess.add_part(arr, n, t)
            locs[t] = loc
    if mode == "whole":
        rows = [len(l) for l in locs]
        my_parts = [(fi, 0, rows[fi]) for fi in range(len(files))]
    if d.world > 1:
        N = d.max_ints([N or 0])[0]                    # a rank without bins learns the width from the others
    if mode == "assigned":
        rows = _exchange_rows(d, my_parts, locs, len(files))
        tm.lap("parse + upload + expected counts (each file once, on one rank)")
        plans = plan_partition(rows, d.world)
        starts = np.concatenate([[0], np.cumsum(rows)])
        mine = {fi: (pids[t], locs[t]) for t, (fi, _lo, _hi) in enumerate(my_parts)}
        pids, locs = _redistribute(d, sess, plans, owner, mine, starts, N)
        my_parts = plans[d.rank]
        tm.lap("hand border pieces to their owners")
    else:
        plans = plan_partition(rows, d.world) if d.world > 1 else [my_parts]
        tm.lap("parse + upload + expected counts")
    if not N:
        N = _columns_of(files[0])
    if hasattr(sess, "pool") and os.environ.get("EPILOGOS_POOL_CLOSE") == "1":
        # every part is uploaded: the staging buffers could be un-locked in the background now.  Measured (gpurun_out/r04ab):
        # 8 GB less to give back at exit (0.6 -> 0.5 s) but STEP 4 and the writers, which fault pages meanwhile, lose more.  Off.
        sess.pool.close()
    sess.ensure_acc(N)
    sess.all_reduce(d)                                # the one exchange step; a rank without bins contributes zeros
    # STEP 2 (identical normalisation on every rank) and STEP 3 of every part enqueued without a host synchronisation; then
    # the host side: count check, S1 table verification, exp_freq as a host array (bench.py times this very sequence)
    sess.launch(int(sum(rows)), N, pids)
    q = sess.finish(int(sum(rows)), N)
    if d.rank == 0:
        np.save(outputDir / "exp_freq_{}.npy".format(fileTag), q, allow_pickle=False)
    tm.lap("all-reduce + check + normalise")

    # STEP 3: scores from the resident data; text is formatted and compressed by writer threads while the next part scores
    payloads = []
    nwriters = min(2, _io.host_budget())               # two files are written at a time: half of this rank's cores each
    wthreads = max(1, _io.host_budget() // nwriters)
    writer = ThreadPoolExecutor(max_workers=nwriters)
    jobs = []
    try:
        for k, (fi, lo, hi) in enumerate(my_parts):
            sc = sess.scores(pids[k])
            whole = lo == 0 and hi == rows[fi]
            name = "scores_{}_{}.txt.gz".format(fileTag, fileStem(files[fi])) if whole else _part_name("scores", fileTag, fi, lo)
            jobs.append(writer.submit(writeScores, sc, outputDir / name, locs[k], None, wthreads))
            payloads.append([sc, np.asarray(locs[k].blob), np.asarray(locs[k].offsets)])
        tm.lap("scores (kernels + download)")
    except BaseException:
        writer.shutdown(wait=True)
        raise
    if defer_writes and d.world == 1:
        # one process: the text goes on being formatted and compressed in the writer threads while the caller runs STEP 4 on the
        # arrays (finish_writes() waits for them); several ranks need their part files complete before rank 0 assembles them
        _deferred.append((writer, jobs, tm, "write text (under STEP 4)"))
    else:
        _wait_writes(writer, jobs)
        tm.lap("write text (overlapped tail)")
    tm.note("H2D uploads: %d for %d part(s)" % (getattr(sess, "n_uploads", 0), len(my_parts)))
    d.barrier()
    got = _gather_parts(d, plans, payloads, 3)
    results = None
    if d.rank == 0:
        results = {}
        for fi, f in enumerate(files):
            stem = fileStem(f)
            owners = sorted((lo, hi) for parts in plans for (pf, lo, hi) in parts if pf == fi)
            _assemble_text(outputDir, "scores", "scores_{}_{}.txt.gz".format(fileTag, stem), fileTag, fi, owners, rows[fi])
            if not owners:                             # an empty input file still gets its (empty) output
                writeScores(np.zeros((0, numStates), dtype=np.float32), outputDir / "scores_{}_{}.txt.gz".format(fileTag, stem),
                            _io.Locations(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64)))
            parts = [got[(fi, lo)] for lo, _ in owners]
            scoreArr = _cat([p[0] for p in parts], np.zeros((0, numStates), dtype=np.float32))
            loc = _cat_locations([_io.Locations(p[1], p[2]) for p in parts])
            chrName = loc.slice(0, 1).to_object_array()[0, 0] if len(loc) else ""
            results[stem] = (chrName, scoreArr, loc)
            if keep_temp_scores:
                np.savez_compressed(outputDir / "temp_scores_{}_{}.npz".format(fileTag, stem), chrName=np.array([chrName]),
                                    scoreArr=scoreArr, locationArr=loc.to_object_array())
        tm.lap("assemble files" + (" + temp_scores npz" if keep_temp_scores else ""))
    _io.log_thread_census("single s%d rank %d of %d" % (saliency, d.rank, d.world))
    d.barrier()
    return q, results


def run_paired_groups(files1, files2, numStates, saliency, outputDir, fileTag, quiescentState, groupSize, nullSeed,
                      verbose=False, backend=None, device=None, keep_temps=True, defer_writes=False):
    """STEP 1-3 of paired mode (reference run.py:205-221,258-279 + scores.py:172-256) over the bin-range partition.
    Background counts are taken over the column concatenation [A|B] (helpers.py:173) -- from the two groups' own
    histograms, each group is uploaded once -- all-reduced once; each rank then scores A, B and the two shuffled null
    groups of its bins from the resident data.  The null shuffle is keyed by (seed, file ordinal, row in the file)
    (shuffle_key), so the outputs do not depend on the number of GPUs.  Returns (exp_freq, results); results (rank 0) maps file stem ->
    dict(chrName, locations, nullDistances, quiescenceArr, distances, maxDiff) for an in-process STEP 4; keep_temps
    also writes temp_nullDistances / temp_quiescence (the reference's, scores.py:246-255) and temp_pairMetrics (the
    side-car of this engine's STEP 4)."""
    be = backend if backend is not None else _backend.get()
    _io.set_state_limit(numStates)
    d = _Dist()
    files1, files2 = [Path(f) for f in files1], [Path(f) for f in files2]
    outputDir = Path(outputDir)
    tm = _Timer(d.rank)
    if d.rank == 0:
        _clean_parts(outputDir, "pairwiseDelta", fileTag)
    d.barrier()
    mode, rows, my_parts, owner = _plan(files1, d, tm, files2)
    sess = _open_paired(be, numStates, saliency, quiescentState, groupSize, nullSeed)

    # group 1 and group 2 of a part are jobs 2k and 2k + 1; the second group follows the first one's row ranges
    jobs = []
    for fi, lo, hi in my_parts:
        jobs += [(files1[fi], lo, hi), (files2[fi], lo, hi)]
    locs, pids, NA, NB = [None] * len(my_parts), [None] * len(my_parts), None, None
    halves = {}                                        # part index -> its group that arrived first, already in HBM
    with closing(_stream_parts(jobs, sess, numStates)) as stream:
        for t, x, n, loc in stream:                    # in order of completion; jobs 2k and 2k + 1 are part k
            k, second = t // 2, t % 2
            if second:
                NB = max(NB or 0, n)                   # (an empty file has no width)
            else:
                NA, locs[k] = max(NA or 0, n), loc
            staged = (sess.stage(x, n, t), n, x.shape[0])
            if k not in halves:
                halves[k] = (second, staged)
                continue
            other_second, other = halves.pop(k)
            (XA, nA, rA), (XB, nB, rB) = (other, staged) if second else (staged, other)
            fi, lo, hi = my_parts[k]
            if rA != rB:
                raise ValueError("paired inputs differ in length: {} vs {}".format(files1[fi], files2[fi]))
            # the key of the part's first row in the null shuffle: (file, row in the file) -- known here in every mode, so the
            # session draws the null groups right behind the count pass (on its second stream)
            pids[k] = sess.add_staged(XA, nA, XB, nB, shuffle_key(fi, lo))
    if mode == "whole":                                # single rank: the row counts come from the parse
        rows = [len(l) for l in locs]
        my_parts = [(fi, 0, rows[fi]) for fi in range(len(files1))]
    if d.world > 1:
        NA, NB = d.max_ints([NA or 0, NB or 0])
    if mode == "assigned":
        rows = _exchange_rows(d, my_parts, locs, len(files1))
        tm.lap("parse + upload + expected counts (each file once, on one rank)")
        plans = plan_partition(rows, d.world)
        starts = np.concatenate([[0], np.cumsum(rows)])
        mine = {fi: (pids[k], locs[k]) for k, (fi, _lo, _hi) in enumerate(my_parts)}
        pids, locs = _redistribute(d, sess, plans, owner, mine, starts, (NA, NB))
        my_parts = plans[d.rank]
        tm.lap("hand border pieces to their owners")
    else:
        plans = plan_partition(rows, d.world) if d.world > 1 else [my_parts]
        tm.lap("parse + upload + expected counts")
    if not NA:
        NA, NB = _columns_of(files1[0]), _columns_of(files2[0])
    if hasattr(sess, "pool") and os.environ.get("EPILOGOS_POOL_CLOSE") == "1":
        sess.pool.close()
    sess.ensure_acc(NA + NB)
    sess.all_reduce(d)
    sess.launch(int(sum(rows)), NA + NB, pids)        # STEP 2 + the one-launch score pass of all parts, no host sync
    q = sess.finish(int(sum(rows)), NA + NB)
    if d.rank == 0:
        np.save(outputDir / "exp_freq_{}.npy".format(fileTag), q, allow_pickle=False)
    tm.lap("all-reduce + check + normalise")

    payloads = []
    nwriters = min(2, _io.host_budget())
    wthreads = max(1, _io.host_budget() // nwriters)
    writer = ThreadPoolExecutor(max_workers=nwriters)
    wjobs = []
    try:
        for k, (fi, lo, hi) in enumerate(my_parts):
            res = sess.results(pids[k])
            whole = lo == 0 and hi == rows[fi]
            name = ("pairwiseDelta_{}_{}.txt.gz".format(fileTag, fileStem(files1[fi])) if whole
                    else _part_name("pairwiseDelta", fileTag, fi, lo))
            wjobs.append(writer.submit(writeScores, res["delta"], outputDir / name, locs[k], None, wthreads))
            payloads.append([np.asarray(res["null"], dtype=np.float32), np.asarray(res["quies"], dtype=np.bool_),
                             np.asarray(res["rdist"], dtype=np.float32), np.asarray(res["mdiff"], dtype=np.int32),
                             np.asarray(locs[k].blob), np.asarray(locs[k].offsets)])
    except BaseException:
        writer.shutdown(wait=True)
        raise
    if defer_writes and d.world == 1:
        tm.lap("scores, nulls, deltas (kernels + download)")
        _deferred.append((writer, wjobs, tm, "write text (under STEP 4)"))
    else:
        _wait_writes(writer, wjobs)
        tm.lap("scores, nulls, deltas + write text")
    tm.note("H2D uploads: %d for %d part(s) x 2 groups" % (getattr(sess, "n_uploads", 0), len(my_parts)))
    d.barrier()
    got = _gather_parts(d, plans, payloads, 6)
    results = None
    if d.rank == 0:
        results = {}
        for fi, f in enumerate(files1):
            stem = fileStem(f)
            owners = sorted((lo, hi) for parts in plans for (pf, lo, hi) in parts if pf == fi)
            _assemble_text(outputDir, "pairwiseDelta", "pairwiseDelta_{}_{}.txt.gz".format(fileTag, stem), fileTag, fi, owners, rows[fi])
            if not owners:
                writeScores(np.zeros((0, numStates), dtype=np.float32), outputDir / "pairwiseDelta_{}_{}.txt.gz".format(fileTag, stem),
                            _io.Locations(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64)))
            parts = [got[(fi, lo)] for lo, _ in owners]
            loc = _cat_locations([_io.Locations(p[4], p[5]) for p in parts])
            chrName = loc.slice(0, 1).to_object_array()[0, 0] if len(loc) else ""
            res = {"chrName": chrName, "locations": loc,
                   "nullDistances": _cat([p[0] for p in parts], np.zeros(0, dtype=np.float32)).astype(np.float32),
                   "quiescenceArr": _cat([p[1] for p in parts], np.zeros(0, dtype=bool)).astype(bool),
                   "distances": _cat([p[2] for p in parts], np.zeros(0, dtype=np.float32)).astype(np.float32),
                   "maxDiff": _cat([p[3] for p in parts], np.zeros(0, dtype=np.int32)).astype(np.int32)}
            results[stem] = res
            if keep_temps:
                np.savez_compressed(outputDir / "temp_nullDistances_{}_{}.npz".format(fileTag, stem), chrName=np.array([chrName]),
                                    nullDistances=res["nullDistances"])
                np.savez_compressed(outputDir / "temp_quiescence_{}_{}.npz".format(fileTag, stem), chrName=np.array([chrName]),
                                    quiescenceArr=res["quiescenceArr"])
                # side-car for this engine's STEP 4 (roiAndVisualPairwise.readInData): the per-bin reduction the
                # reference redoes from the pairwiseDelta text, already computed on the GPU; removed with the other temps
                starts, ends = loc.start_end()
                np.savez_compressed(outputDir / "temp_pairMetrics_{}_{}.npz".format(fileTag, stem), chrName=np.array([chrName]),
                                    distances=res["distances"], maxDiff=res["maxDiff"], starts=starts, ends=ends)
        tm.lap("assemble files" + (" + temp npz" if keep_temps else ""))
    _io.log_thread_census("paired s%d rank %d of %d" % (saliency, d.rank, d.world))
    d.barrier()
    return q, results

"""Genome-wide driver: replaces the reference's per-chromosome SLURM fan-out (run.py:190-279) with a contiguous
bin-range partition across the GPUs of one node and a single all-reduce of the state-count array.

The bins of all input files, concatenated in file order, are split with the reference's own splitRows rule
(helpers.py:116-118): rank g of G owns global bins [g*R//G, (g+1)*R//G).  Pass 1 counts locally, one
all-reduce(SUM) over torch.distributed (backend nccl = RCCL over xGMI on GPUs; gloo in the CPU tests) makes the
counts global, every rank normalises identically, pass 2 scores locally with no further communication.  Integer
sums make the result independent of G and of the reduction order.  Outputs keep the reference's names: every rank
writes gzip members for its bin ranges and rank 0 concatenates them per chromosome (a multi-member gzip file is
a valid gzip file)."""
import os
import shutil
from pathlib import Path

import numpy as np

from . import _io
from . import backend as _backend
from .helpers import countRows, fileStem, readLocations, readStates, readTable, splitRows
from .scores import writeScores


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


class _Dist:
    """Thin wrapper so that the single-process case needs no process group."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1

    def all_reduce_counts(self, counts, device=None):
        """SUM-all-reduce an integer numpy array; on GPUs the tensor lives on the device so RCCL moves it over xGMI."""
        if not self.dist:
            return counts
        import torch
        t = torch.from_numpy(np.ascontiguousarray(counts))
        if device is not None:
            t = t.to(device)
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


def _count_rows(files):
    """countRows of every file (a gunzip pass each), files in parallel: the native counter releases the GIL."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(16, len(files)))) as pool:
        return list(pool.map(countRows, files))


def _read_parts(files, parts):
    """readTable of every (file, lo, hi) of this rank, files in parallel (gz inflate is serial per file); hi = None
    reads to the last complete line, which is what countRows counts (helpers.py:94)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, len(parts)))) as pool:
        return list(pool.map(lambda part: readTable(files[part[0]], None if part[2] is None else (part[1], part[2])), parts))


def _plan_and_read(files, d, tm):
    """(rows per file, this rank's parts, their parsed tables).  A single rank reads whole files and takes the row counts
    from the parse instead of a separate gunzip pass per file."""
    if d.world == 1:
        tables = _read_parts(files, [(fi, 0, None) for fi in range(len(files))])
        rows = [t[0].shape[0] for t in tables]
        tm.lap("parse (row counts included)")
        return rows, [(fi, 0, rows[fi]) for fi in range(len(files))], tables
    rows = _count_rows(files)
    tm.lap("count rows")
    parts = plan_partition(rows, d.world)[d.rank]
    tables = _read_parts(files, parts)
    tm.lap("parse")
    return rows, parts, tables


def _publish(outputDir, kind, fileTag, stem, lo, rank, payload, mine):
    """Hand a part's arrays to rank 0: kept in memory on rank 0 itself, an uncompressed .npz part file otherwise."""
    if rank == 0:
        mine.setdefault(stem, []).append((lo, payload))
    else:
        np.savez(outputDir / ".part_{}_{}_{}_{:012d}.npz".format(kind, fileTag, stem, lo), **payload)


def _collect(outputDir, kind, fileTag, stem, mine):
    """All parts of one file in bin order (rank 0): its own from memory, the other ranks' from their part files."""
    parts = list(mine.get(stem, []))
    prefix = ".part_{}_{}_{}_".format(kind, fileTag, stem)
    for p in outputDir.glob(prefix + "*.npz"):
        with np.load(p) as z:
            parts.append((int(p.name[len(prefix):-4]), {k: z[k] for k in z.files}))
        os.remove(p)
    parts.sort(key=lambda t: t[0])
    return [pl for _, pl in parts]


def _cat(arrs, empty):
    return np.concatenate(arrs) if arrs else empty


def _cat_locations(parts):
    blobs = [pl["loc_blob"] for pl in parts]
    offs, base = [np.zeros(1, dtype=np.int64)], 0
    for pl in parts:
        offs.append(pl["loc_off"][1:] + base)
        base += int(pl["loc_off"][-1])
    return _io.Locations(_cat(blobs, np.zeros(0, dtype=np.uint8)), np.concatenate(offs))


def _cat_gzip_members(outputDir, name, pattern):
    with open(outputDir / name, "wb") as out:
        for p in sorted(outputDir.glob(pattern)):
            with open(p, "rb") as src:
                shutil.copyfileobj(src, out)
            os.remove(p)


def run_single_group(files, numStates, saliency, outputDir, fileTag, verbose=False, backend=None, device=None,
                     keep_temp_scores=True):
    """STEP 1-3 for a single group over `files` (one per chromosome).  Returns (exp_freq float32, results) where results
    (rank 0 only, else None) maps file stem -> (chrName, float32 scores [R, S], _io.Locations) for an in-process STEP 4.
    keep_temp_scores writes the reference's temp_scores_{tag}_{stem}.npz (scores.py:166-169) for a STEP 4 run
    elsewhere; the command line skips them because its STEP 4 would delete them a moment later."""
    be = backend if backend is not None else _backend.get()
    d = _Dist()
    files = [Path(f) for f in files]
    outputDir = Path(outputDir)
    tm = _Timer(d.rank)
    rows, my_parts, tables = _plan_and_read(files, d, tm)

    # STEP 1: local counts over my bin ranges
    N = tables[0][0].shape[1] if tables else readStates(file1Path=files[0], rowsToCalc=(0, 1), verbose=False).shape[1]
    shape = {1: (numStates,), 2: (numStates, numStates), 3: (N, N, numStates, numStates)}[saliency]
    if hasattr(be, "counts_begin"):              # product backend: counts stay in HBM through all-reduce and normalise
        acc = be.counts_begin(numStates, saliency, N)
        for x, _ in tables:
            be.counts_add(acc, x, numStates, saliency)
        tm.lap("expected counts")
        d.all_reduce_tensor(acc)                 # the one exchange step; a rank without bins contributes zeros
        q_score, q = be.counts_finish(acc, shape)
        del acc
    else:
        counts = np.zeros(shape, dtype=np.int32 if saliency == 3 else np.int64)
        for x, _ in tables:
            counts = counts + be.expected_counts(x, numStates, saliency)
        tm.lap("expected counts")
        counts = d.all_reduce_counts(counts, device=device)
        q = be.normalise(counts)                 # STEP 2: identical normalisation on every rank
        q_score = q
    if d.rank == 0:
        np.save(outputDir / "exp_freq_{}.npy".format(fileTag), q, allow_pickle=False)
    tm.lap("all-reduce + normalise")

    # STEP 3: local scores, written as gzip members per (file, range)
    mine = {}
    t_sc = t_wr = 0.0
    for (fi, lo, hi), (x, loc) in zip(my_parts, tables):
        t0 = tm.time()
        sc = be.scores(x, numStates, saliency, q_score)
        t1 = tm.time()
        stem = fileStem(files[fi])
        writeScores(sc, outputDir / ".part_scores_{}_{}_{:012d}.gz".format(fileTag, stem, lo), loc)
        _publish(outputDir, "scores", fileTag, stem, lo, d.rank, {"scores": sc, "loc_blob": loc.blob, "loc_off": loc.offsets}, mine)
        t_sc += t1 - t0
        t_wr += tm.time() - t1
    if tm.on:
        print("    [timing] %-34s %7.2f s\n    [timing] %-34s %7.2f s" % ("scores (upload + kernels + download)", t_sc, "write text", t_wr), flush=True)
        tm.t = tm.time()
    d.barrier()
    results = None
    if d.rank == 0:
        results = {}
        for f in files:
            stem = fileStem(f)
            _cat_gzip_members(outputDir, "scores_{}_{}.txt.gz".format(fileTag, stem), ".part_scores_{}_{}_*.gz".format(fileTag, stem))
            parts = _collect(outputDir, "scores", fileTag, stem, mine)
            scoreArr = _cat([pl["scores"] for pl in parts], np.zeros((0, numStates), dtype=np.float32))
            loc = _cat_locations(parts)
            chrName = loc.slice(0, 1).to_object_array()[0, 0] if len(loc) else ""
            results[stem] = (chrName, scoreArr, loc)
            if keep_temp_scores:
                np.savez_compressed(outputDir / "temp_scores_{}_{}.npz".format(fileTag, stem), chrName=np.array([chrName]),
                                    scoreArr=scoreArr, locationArr=loc.to_object_array())
        tm.lap("assemble files" + (" + temp_scores npz" if keep_temp_scores else ""))
    d.barrier()
    return q, results


def run_paired_groups(files1, files2, numStates, saliency, outputDir, fileTag, quiescentState, groupSize, nullSeed,
                      verbose=False, backend=None, device=None, keep_temps=True):
    """STEP 1-3 of paired mode (reference run.py:205-221,258-279 + scores.py:172-256) over the bin-range partition.
    Background counts are taken over the column concatenation [A|B] (helpers.py:173), all-reduced once; each rank then
    scores A, B and the two shuffled null groups of its bins.  The null shuffle is keyed by (seed, global bin index), so
    the outputs do not depend on the number of GPUs.  Returns (exp_freq, results); results (rank 0) maps file stem ->
    dict(chrName, locations, nullDistances, quiescenceArr, distances, maxDiff) for an in-process STEP 4; keep_temps
    also writes temp_nullDistances / temp_quiescence (the reference's, scores.py:246-255) and temp_pairMetrics (the
    side-car of this engine's STEP 4)."""
    be = backend if backend is not None else _backend.get()
    d = _Dist()
    files1, files2 = [Path(f) for f in files1], [Path(f) for f in files2]
    outputDir = Path(outputDir)
    tm = _Timer(d.rank)
    rows, my_parts, ta = _plan_and_read(files1, d, tm)
    file_start = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    tb = _read_parts(files2, my_parts)                       # the second group follows the first one's row ranges
    tm.lap("parse group 2")
    counts, chunks = None, []
    for (fi, lo, hi), (xa, loc), (xb, _) in zip(my_parts, ta, tb):
        if xb.shape[0] != xa.shape[0]:
            raise ValueError("paired inputs differ in length: {} vs {}".format(files1[fi], files2[fi]))
        chunks.append((xa, xb, loc))
        c = be.expected_counts(np.concatenate((xa, xb), axis=1), numStates, saliency)
        counts = c if counts is None else counts + c
    del ta, tb
    if counts is None:
        counts = np.zeros((numStates,) if saliency == 1 else (numStates, numStates), dtype=np.int64)
    counts = d.all_reduce_counts(counts, device=device)
    q = be.normalise(counts)
    if d.rank == 0:
        np.save(outputDir / "exp_freq_{}.npy".format(fileTag), q, allow_pickle=False)
    tm.lap("expected counts + all-reduce")

    mine = {}
    for (fi, lo, hi), (xa, xb, loc) in zip(my_parts, chunks):
        n1, n2 = xa.shape[1], xb.shape[1]
        s1 = be.scores(xa, numStates, saliency, q, perms=n1 * (n1 - 1))
        s2 = be.scores(xb, numStates, saliency, q, perms=n2 * (n2 - 1))
        na, nb = be.null_scores(xa, xb, numStates, saliency, q, groupSize, nullSeed, row0=int(file_start[fi] + lo))
        delta, _ = be.pair_finish(s1, s2)
        _, dist = be.pair_finish(na, nb)
        real_dist, maxdiff = be.pair_metrics(delta, roundtrip=True)      # what STEP 4 would recompute from the text
        quies = be.quiescent(xa, xb, quiescentState)
        stem = fileStem(files1[fi])
        writeScores(delta, outputDir / ".part_pairwiseDelta_{}_{}_{:012d}.gz".format(fileTag, stem, lo), loc)
        _publish(outputDir, "pair", fileTag, stem, lo, d.rank,
                 {"null": dist, "quies": quies, "rdist": real_dist, "mdiff": maxdiff, "loc_blob": loc.blob, "loc_off": loc.offsets}, mine)
    tm.lap("scores, nulls, deltas + write text")
    d.barrier()
    results = None
    if d.rank == 0:
        results = {}
        for f in files1:
            stem = fileStem(f)
            _cat_gzip_members(outputDir, "pairwiseDelta_{}_{}.txt.gz".format(fileTag, stem),
                              ".part_pairwiseDelta_{}_{}_*.gz".format(fileTag, stem))
            parts = _collect(outputDir, "pair", fileTag, stem, mine)
            loc = _cat_locations(parts)
            chrName = loc.slice(0, 1).to_object_array()[0, 0] if len(loc) else ""
            res = {"chrName": chrName, "locations": loc,
                   "nullDistances": _cat([pl["null"] for pl in parts], np.zeros(0, dtype=np.float32)),
                   "quiescenceArr": _cat([pl["quies"] for pl in parts], np.zeros(0, dtype=bool)).astype(bool),
                   "distances": _cat([pl["rdist"] for pl in parts], np.zeros(0, dtype=np.float32)).astype(np.float32),
                   "maxDiff": _cat([pl["mdiff"] for pl in parts], np.zeros(0, dtype=np.int32)).astype(np.int32)}
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
    d.barrier()
    return q, results

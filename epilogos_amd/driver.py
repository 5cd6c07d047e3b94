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

    def barrier(self):
        if self.dist:
            self.dist.barrier()


def run_single_group(files, numStates, saliency, outputDir, fileTag, verbose=False, backend=None, device=None):
    """STEP 1-3 for a single group over `files` (one per chromosome).  Returns exp_freq (float32)."""
    be = backend if backend is not None else _backend.get()
    d = _Dist()
    files = [Path(f) for f in files]
    outputDir = Path(outputDir)
    rows = [countRows(f) for f in files]
    my_parts = plan_partition(rows, d.world)[d.rank]

    # STEP 1: local counts over my bin ranges
    counts, chunks, locs = None, [], []
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, len(my_parts)))) as pool:   # gz inflate is serial per file
        tables = list(pool.map(lambda part: readTable(files[part[0]], (part[1], part[2])), my_parts))
    for (fi, lo, hi), (x, loc) in zip(my_parts, tables):
        chunks.append(x)
        locs.append(loc)
        c = be.expected_counts(x, numStates, saliency)
        counts = c if counts is None else counts + c
    if counts is None:   # a rank without bins still takes part in the collective
        N = readStates(file1Path=files[0], rowsToCalc=(0, 1), verbose=False).shape[1]
        shape = {1: (numStates,), 2: (numStates, numStates), 3: (N, N, numStates, numStates)}[saliency]
        counts = np.zeros(shape, dtype=np.int32 if saliency == 3 else np.int64)
    # the one exchange step
    counts = d.all_reduce_counts(counts, device=device)
    # STEP 2: identical normalisation on every rank
    q = be.normalise(counts)
    if d.rank == 0:
        np.save(outputDir / "exp_freq_{}.npy".format(fileTag), q, allow_pickle=False)

    # STEP 3: local scores, written as gzip members per (file, range)
    for (fi, lo, hi), x, loc in zip(my_parts, chunks, locs):
        sc = be.scores(x, numStates, saliency, q)
        stem = fileStem(files[fi])
        part = outputDir / ".part_scores_{}_{}_{:012d}.gz".format(fileTag, stem, lo)
        writeScores(sc, part, loc)
        np.save(outputDir / ".part_scores_{}_{}_{:012d}.npy".format(fileTag, stem, lo), sc, allow_pickle=False)
    d.barrier()
    if d.rank == 0:
        for f in files:
            stem = fileStem(f)
            parts = sorted(outputDir.glob(".part_scores_{}_{}_*.gz".format(fileTag, stem)))
            with open(outputDir / "scores_{}_{}.txt.gz".format(fileTag, stem), "wb") as out:
                for p in parts:
                    with open(p, "rb") as src:
                        shutil.copyfileobj(src, out)
                    os.remove(p)
            arrs = []
            for p in sorted(outputDir.glob(".part_scores_{}_{}_*.npy".format(fileTag, stem))):
                arrs.append(np.load(p))
                os.remove(p)
            loc = readLocations(f)[:sum(a.shape[0] for a in arrs)]
            np.savez_compressed(outputDir / "temp_scores_{}_{}.npz".format(fileTag, stem), chrName=np.array([loc[0, 0]]),
                                scoreArr=np.concatenate(arrs, axis=0), locationArr=loc)
    d.barrier()
    return q


def run_paired_groups(files1, files2, numStates, saliency, outputDir, fileTag, quiescentState, groupSize, nullSeed,
                      verbose=False, backend=None, device=None):
    """STEP 1-3 of paired mode (reference run.py:205-221,258-279 + scores.py:172-256) over the bin-range partition.
    Background counts are taken over the column concatenation [A|B] (helpers.py:173), all-reduced once; each rank then
    scores A, B and the two shuffled null groups of its bins.  The null shuffle is keyed by (seed, global bin index), so
    the outputs do not depend on the number of GPUs."""
    be = backend if backend is not None else _backend.get()
    d = _Dist()
    files1, files2 = [Path(f) for f in files1], [Path(f) for f in files2]
    outputDir = Path(outputDir)
    rows = [countRows(f) for f in files1]
    my_parts = plan_partition(rows, d.world)[d.rank]
    file_start = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)

    counts, chunks = None, []
    for (fi, lo, hi) in my_parts:
        xa, loc = readTable(files1[fi], (lo, hi))
        xb, _ = readTable(files2[fi], (lo, hi))
        if xb.shape[0] != xa.shape[0]:
            raise ValueError("paired inputs differ in length: {} vs {}".format(files1[fi], files2[fi]))
        chunks.append((xa, xb, loc))
        c = be.expected_counts(np.concatenate((xa, xb), axis=1), numStates, saliency)
        counts = c if counts is None else counts + c
    if counts is None:
        counts = np.zeros((numStates,) if saliency == 1 else (numStates, numStates), dtype=np.int64)
    counts = d.all_reduce_counts(counts, device=device)
    q = be.normalise(counts)
    if d.rank == 0:
        np.save(outputDir / "exp_freq_{}.npy".format(fileTag), q, allow_pickle=False)

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
        np.save(outputDir / ".part_null_{}_{}_{:012d}.npy".format(fileTag, stem, lo), dist, allow_pickle=False)
        np.save(outputDir / ".part_quies_{}_{}_{:012d}.npy".format(fileTag, stem, lo), quies, allow_pickle=False)
        np.save(outputDir / ".part_rdist_{}_{}_{:012d}.npy".format(fileTag, stem, lo), real_dist, allow_pickle=False)
        np.save(outputDir / ".part_mdiff_{}_{}_{:012d}.npy".format(fileTag, stem, lo), maxdiff, allow_pickle=False)
    d.barrier()
    if d.rank == 0:
        for f in files1:
            stem = fileStem(f)
            with open(outputDir / "pairwiseDelta_{}_{}.txt.gz".format(fileTag, stem), "wb") as out:
                for p in sorted(outputDir.glob(".part_pairwiseDelta_{}_{}_*.gz".format(fileTag, stem))):
                    with open(p, "rb") as src:
                        shutil.copyfileobj(src, out)
                    os.remove(p)
            parts = {}
            for kind in ("null", "quies", "rdist", "mdiff"):
                arrs = []
                for p in sorted(outputDir.glob(".part_{}_{}_{}_*.npy".format(kind, fileTag, stem))):
                    arrs.append(np.load(p))
                    os.remove(p)
                parts[kind] = np.concatenate(arrs) if arrs else np.zeros(0)
            locs = readLocations(f)
            chrName = locs[0, 0]
            np.savez_compressed(outputDir / "temp_nullDistances_{}_{}.npz".format(fileTag, stem), chrName=np.array([chrName]),
                                nullDistances=parts["null"])
            np.savez_compressed(outputDir / "temp_quiescence_{}_{}.npz".format(fileTag, stem), chrName=np.array([chrName]),
                                quiescenceArr=parts["quies"].astype(bool))
            # side-car for this engine's STEP 4 (roiAndVisualPairwise.readInData): the per-bin reduction the reference
            # redoes from the pairwiseDelta text, already computed on the GPU; removed with the other temp_*.npz
            n = len(parts["rdist"])
            np.savez_compressed(outputDir / "temp_pairMetrics_{}_{}.npz".format(fileTag, stem), chrName=np.array([chrName]),
                                distances=parts["rdist"].astype(np.float32), maxDiff=parts["mdiff"].astype(np.int32),
                                starts=locs[:n, 1].astype(np.int64), ends=locs[:n, 2].astype(np.int64))
    d.barrier()
    return q

"""STEP 4 of a single-group run -- regions of interest.  Same signature, inputs and output file as the
reference's epilogos/roiSingle.py (main :10-40, readInData :43-76, createTopScoresTxt :95-142): reads the
temp_scores_*.npz of STEP 3 in chromosome order, picks the top 100 non-overlapping windows with the "maxmean" rule of
the reference's vendored filter_regions (filter_regions.py:375-448 through helpers.maxMean :253-274) and writes
regionsOfInterest_{fileTag}.txt; then removes the temporaries and exp_freq like the reference (quirk Q4).

The selection is a sequential greedy pick on a 1-D vector: it stays on the host.  Rolling max / mean use the same pandas
primitive the reference calls, so that ties between windows that share their maximum are broken by bit-identical means."""
from os import remove
from pathlib import Path
from sys import argv
from time import time

import numpy as np
import pandas as pd

from . import _io
from .helpers import strToBool


def getStateNames(stateFile):
    """Shorthand state names (reference helpers.py:20-28)."""
    return pd.read_table(Path(stateFile), header=0, sep="\t")["short_name"].values


def orderChromosomes(chromosomes):
    """Numbered chromosomes ascending, then the others alphabetically (reference helpers.py:224-250)."""
    ints, strs = [], []
    for c in chromosomes:
        tail = c.split("chr")[-1]
        try:
            ints.append(int(tail))
        except ValueError:
            strs.append(tail)
    return ["chr" + str(t) for t in sorted(ints) + sorted(strs)]


def findSign(x):
    return "+" if x >= 0 else "-"


def maxMean(chrom, start, end, score, roiWidth, maxRegions, _first_candidates=None):
    """Top `maxRegions` non-overlapping windows of `roiWidth` bins by (rolling max, rolling mean, centre score).
    Restates filter_regions.Filter.maxmean (:375-448) + helpers.maxMean (:253-274) on plain arrays.
    Returns (chromosome, window start, window end, score = rolling max, original centre index), best first."""
    W, h = int(roiWidth), int(roiWidth) // 2
    R = len(score)
    score = np.asarray(score, dtype=np.float64)
    # window coordinates: Start of the bin h to the left, End of the bin h (odd W) / h-1 (even W) to the right
    e_off = h if W % 2 else h - 1
    lo, hi = h, R - e_off                      # rows that have both shifted coordinates (Series.shift + dropna)
    if hi <= lo:
        return [np.array([])] * 5
    orig = np.arange(lo, hi)
    w_start = np.asarray(start)[orig - h]
    w_end = np.asarray(end)[orig + e_off]
    sc = score[lo:hi]
    # the rolling MEAN stays pandas' own (an online add/remove sum with compensation: its last bits depend on the whole history,
    # and they break ties); the rolling MAX is exact in any implementation -- the native one is pandas' deque, threaded
    rmean = pd.Series(sc).rolling(W, center=True).mean().to_numpy()
    rmax = _io.rolling_max(sc, W)
    ok = ~np.isnan(rmax)                       # incomplete edge windows
    ok &= ~(np.asarray(w_start, dtype=np.int64) >= np.asarray(w_end, dtype=np.int64))   # windows spanning two chromosomes
    keep = np.nonzero(ok)[0]
    orig, w_start, w_end, sc, rmax, rmean = orig[keep], w_start[keep], w_end[keep], sc[keep], rmax[keep], rmean[keep]
    n = len(keep)
    # descending by (max, mean, score); ties keep genomic order (pandas' multi-key sort is stable).  The greedy pick only
    # consumes a PREFIX of that order, so instead of sorting all n windows (a whole genome has 15 M) the windows whose
    # rolling max reaches the m-th largest are sorted -- every one of them, ties at the threshold included, so their order
    # is exactly the prefix of the full sort -- and the candidate set grows only if the pick runs through it.
    m_try = min(n, _first_candidates if _first_candidates else max(4096, 64 * W * maxRegions))
    while True:
        if m_try >= n:
            cand = np.arange(n)
        else:
            thresh = np.partition(rmax, n - m_try)[n - m_try]
            cand = np.nonzero(rmax >= thresh)[0]          # ascending = genomic order, as a stable sort needs
        order = cand[np.lexsort((-sc[cand], -rmean[cand], -rmax[cand]))]
        hits = np.zeros(n, dtype=bool)
        chosen = []
        for m in order:
            if len(chosen) >= maxRegions:
                break
            a = max(m - h, 0)
            b = min(m + h + 1 if W % 2 else m + h, n)
            if not hits[a:b].any():
                hits[a:b] = True
                chosen.append(m)
        if len(chosen) >= maxRegions or len(cand) == n:
            break
        m_try = min(n, 8 * m_try)
    chosen = np.array(sorted(chosen), dtype=np.int64)        # back to genomic order, then best first (stable)
    # helpers.py:272 sorts by [RollingMax, RollingMean, Score], but by then Filter.filter has overwritten Score with RollingMax
    # (filter_regions.py:215-216, aggregation "max"): the third key repeats the first, windows that tie on max and mean stay
    # in genomic order (pinned by tests/golden/roi.npz roi_tie_*; sorting by the centre score instead does NOT match)
    final = chosen[np.lexsort((-rmean[chosen], -rmax[chosen]))]
    return (np.asarray(chrom)[orig[final]], w_start[final], w_end[final], rmax[final], orig[final])


def readInData(outputDirPath):
    """All temp_scores_*.npz in chromosome order (reference roiSingle.py:43-76); removes them afterwards."""
    chunks = {}
    for file in Path(outputDirPath).glob("temp_scores_*.npz"):
        z = np.load(file, allow_pickle=True)
        chunks[z["chrName"][0]] = (z["scoreArr"], z["locationArr"])
    order = orderChromosomes(list(chunks))
    scoreArr = np.concatenate([chunks[c][0] for c in order])
    locationArr = np.concatenate([chunks[c][1] for c in order])
    for file in Path(outputDirPath).glob("temp_scores_*.npz"):
        remove(file)
    return locationArr, scoreArr


def createTopScoresTxt(filePath, locationArr, scoreArr, nameArr, roiWidth):
    """regionsOfInterest*.txt: chromosome, start, end, largest-scoring state, |sum of scores|, sign
    (reference roiSingle.py:95-142)."""
    W = int(roiWidth)
    # scoreArr: one [R, S] array, or the per-chromosome arrays in genomic order (a whole genome's scores are 1.1 GB: they are
    # not concatenated, only their row sums are)
    parts = list(scoreArr) if isinstance(scoreArr, (list, tuple)) else [scoreArr]
    total = np.concatenate([_io.row_sums(a) for a in parts])      # float32 with numpy's own rounding (scoreArr.sum(axis=1))
    bounds = np.cumsum([0] + [len(a) for a in parts])

    def rows(lo, hi):                                             # scoreArr[lo:hi] of the concatenation
        k0, k1 = np.searchsorted(bounds, lo, side="right") - 1, np.searchsorted(bounds, max(hi - 1, lo), side="right") - 1
        return np.concatenate([parts[k][max(lo - bounds[k], 0):min(hi - bounds[k], len(parts[k]))] for k in range(k0, k1 + 1)])

    # locationArr: the reference's [R, 3] object array, or (chromosome, start, end) columns
    c0, c1, c2 = locationArr if isinstance(locationArr, tuple) else (locationArr[:, 0], locationArr[:, 1], locationArr[:, 2])
    chrom, w_start, w_end, sc, centre = maxMean(c0, c1, c2, total, W, 100)
    S = parts[0].shape[1]
    lines = []
    for k in range(len(centre)):
        lo = centre[k] - W // 2
        hi = centre[k] + W // 2 + (1 if W % 2 else 0)
        win = rows(int(lo), int(hi))
        # the state with the largest value anywhere in the window; ties go to the higher state number
        state = S - int(np.argmax(np.max(win[:, ::-1], axis=0)))
        v = float(np.float32(sc[k]))
        lines.append("{}\t{}\t{}\t{}\t{:.5f}\t{}\n".format(chrom[k], int(w_start[k]), int(w_end[k]), nameArr[state - 1], abs(v),
                                                          findSign(v)))
    with open(filePath, "w") as outFile:
        outFile.write("".join(lines))


def mainFromArrays(results, outputDir, stateInfo, fileTag, expFreqPath, roiWidth, verbose):
    """main() on the arrays driver.run_single_group hands back -- {stem: (chrName, scoreArr, _io.Locations)} -- instead
    of temp_scores_*.npz files: same output file, no 170 B/bin round trip through compressed pickles."""
    byChr = {v[0]: v for v in results.values()}
    order = orderChromosomes(list(byChr))
    scoreArr = [byChr[c][1] for c in order]                    # per chromosome, not concatenated (createTopScoresTxt)
    cols = [byChr[c][2].columns() for c in order]              # native parse of the verbatim "chr\tstart\tend" text
    locationArr = tuple(np.concatenate([c[k] for c in cols]) for k in range(3))
    if not verbose: print("    Regions of interest txt\t", end="", flush=True)
    createTopScoresTxt(Path(outputDir) / "regionsOfInterest_{}.txt".format(fileTag), locationArr, scoreArr,
                       getStateNames(stateInfo), roiWidth)
    if not verbose: print("\t[Done]", flush=True)
    remove(Path(expFreqPath))


def main(outputDir, stateInfo, fileTag, expFreqPath, roiWidth, verbose):
    outputDirPath = Path(outputDir)
    stateNameList = getStateNames(stateInfo)
    if verbose: tRead = time()
    else: print("    Reading in files\t", end="", flush=True)
    locationArr, scoreArr = readInData(outputDirPath)
    print("    Time:", time() - tRead, flush=True) if verbose else print("\t[Done]", flush=True)
    if not verbose: print("    Regions of interest txt\t", end="", flush=True)
    createTopScoresTxt(outputDirPath / "regionsOfInterest_{}.txt".format(fileTag), locationArr, scoreArr, stateNameList,
                       roiWidth)
    if not verbose: print("\t[Done]", flush=True)
    remove(Path(expFreqPath))


if __name__ == "__main__":
    main(argv[1], argv[2], argv[3], argv[4], int(argv[5]), strToBool(argv[6]))

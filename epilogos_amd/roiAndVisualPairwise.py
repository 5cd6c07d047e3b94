"""STEP 4 of a paired run -- p-values / z-scores, metrics and regions of interest.  Same signature and
output files as the reference's epilogos/roiAndVisualPairwise.py main (:20-172) except for the figures:

    pairwiseMetrics_{tag}.txt.gz      writeMetrics :520-573
    regionsOfInterest_{tag}.txt       createROITxt :646-723 (with -n) / createROINoSignificance :726-778
    significantLoci_{tag}.txt.gz      createSignificantLociTxt :576-643 (with -n)

and it removes temp_*.npz and exp_freq like the reference (:341-343, :169).  The Manhattan and diagnostic plots
(:781-1190, matplotlib) are visualisation and are not produced.

The per-bin reduction of readInData (:347-354) -- signed squared distance and largest-difference state of every bin --
runs on the GPU (epg_pair_metrics): the scoring pass leaves it in temp_pairMetrics_{tag}_{stem}.npz next to the other
temporaries; without that side-car (outputs written by the reference itself) pairwiseDelta_*.txt.gz is parsed like the
reference does and the parsed matrix goes through the same kernel.  The statistics stay on the host and call the same
scipy routines as the reference (gennorm fit :245-270, cdf :496-517, zscore :101); the Benjamini-Hochberg step restates
statsmodels.stats.multitest.multipletests(method="fdr_bh") (:93; statsmodels==0.13.2 in requirements.txt, absent from
this image): p_(i) / (i/n) on the ascending p-values, running minimum from the right, clipped at 1."""
import gzip
import warnings
from contextlib import closing
from itertools import repeat
from multiprocessing import Pool, cpu_count
from os import remove
from pathlib import Path
from sys import argv
from time import time

import numpy as np
import pandas as pd
import scipy.stats as st

from . import _io
from . import backend as _backend
from .helpers import getNumStates, strToBool
from .roiSingle import findSign, getStateNames, maxMean, orderChromosomes


def benjaminiHochberg(pvals):
    """fdr_bh adjusted p-values, statsmodels' formula (multitest.py fdrcorrection with method 'indep')."""
    pvals = np.asarray(pvals, dtype=np.float64)
    n = len(pvals)
    if n == 0:
        return pvals.copy()
    order = np.argsort(pvals)
    ecdf = np.arange(1, n + 1) / float(n)
    raw = pvals[order] / ecdf
    corrected = np.minimum.accumulate(raw[::-1])[::-1]
    corrected[corrected > 1] = 1
    out = np.empty_like(corrected)
    out[order] = corrected
    return out


def _ordered_concat(chunks):
    order = orderChromosomes(list(chunks))
    return order, np.concatenate([chunks[c] for c in order]) if order else np.zeros(0)


def readNull(nullFile, quiescenceFile):
    """(chrName, nullDistances), (chrName, quiescenceArr) of one chromosome (reference :224-242)."""
    z, zq = np.load(Path(nullFile)), np.load(Path(quiescenceFile))
    return (z["chrName"][0], z["nullDistances"]), (zq["chrName"][0], zq["quiescenceArr"])


def fitOnSubSample(nullDistances, samplingSize, rng=None):
    """One trial of the null fit (reference roiAndVisualPairwise.py:245-270): generalised-normal parameters fitted to at most
    `samplingSize` null distances drawn without replacement, and the negative log-likelihood of ALL null distances under
    them.  Plain arrays in their own dtype (the float32 the score stage stored: scipy's optimiser sees what the reference's
    sees), (beta, loc, scale) and a float out; every trial draws from fresh OS entropy unless a
    numpy Generator is passed (the reference reseeds the global generator per call, which forked workers need)."""
    values = np.asarray(nullDistances)
    if values.size > samplingSize:
        rng = np.random.default_rng() if rng is None else rng
        fitted_on = values[rng.choice(values.size, size=samplingSize, replace=False)]
    else:
        fitted_on = values
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        params = st.gennorm.fit(fitted_on)
        return params, st.gennorm.nnlf(params, values)


_FIT_DATA = None                                     # the null distances of the pool's parent: workers inherit them through fork


def _fitOnSharedData(samplingSize):
    return fitOnSubSample(_FIT_DATA, samplingSize)


def fitDistances(outputDirPath, numProcesses, numTrials, samplingSize):
    """Null distances of the non-quiescent bins in chromosome order, `numTrials` fits, the one with the median
    negative log-likelihood wins (reference :175-221)."""
    nulls, quies = {}, {}
    for nf, qf in zip(sorted(outputDirPath.glob("temp_nullDistances_*.npz")),
                      sorted(outputDirPath.glob("temp_quiescence_*.npz"))):
        (c, d), (cq, qa) = readNull(nf, qf)
        nulls[c], quies[cq] = d, qa
    _, distanceArrNull = _ordered_concat(nulls)
    _, quiescenceArr = _ordered_concat(quies)
    nonQuiescentIdx = np.where(np.invert(quiescenceArr.astype(bool)))[0]
    return _fitParams(distanceArrNull, quiescenceArr, numProcesses, numTrials, samplingSize), distanceArrNull, nonQuiescentIdx


def readInData(outputDirPath, numStates, backend=None):
    """locationArr int64 [R,3] (chromosome number, start, end), signed squared distances float32 [R], 1-based
    largest-difference state int32 [R] and {number: chromosome}, sorted by location (reference :273-356)."""
    be = backend if backend is not None else _backend.get()
    deltas = sorted(outputDirPath.glob("pairwiseDelta_*.txt.gz"))
    chunks = {}
    for f in deltas:
        tail = f.name[len("pairwiseDelta_"):-len(".txt.gz")]
        side = outputDirPath / "temp_pairMetrics_{}.npz".format(tail)
        if side.exists():
            z = np.load(side)
            chrom = np.full(len(z["distances"]), z["chrName"][0], dtype=object)
            part = (chrom, z["starts"], z["ends"], z["distances"], z["maxDiff"])
        else:
            df = pd.read_table(f, header=None, sep="\t")
            diffArr = np.ascontiguousarray(df.iloc[:, 3:3 + numStates].to_numpy(dtype=np.float32))
            dist, maxdiff = be.pair_metrics(diffArr, roundtrip=False)
            part = (df.iloc[:, 0].to_numpy(dtype=object), df.iloc[:, 1].to_numpy(dtype=np.int64),
                    df.iloc[:, 2].to_numpy(dtype=np.int64), dist, maxdiff)
        chunks[f] = part
    chrom = np.concatenate([c[0] for c in chunks.values()]) if chunks else np.zeros(0, dtype=object)
    chrOrder = orderChromosomes(pd.unique(chrom))
    number = {c: i + 1 for i, c in enumerate(chrOrder)}
    chrDict = {i + 1: c for i, c in enumerate(chrOrder)}
    chrNum = np.array([number[c] for c in chrom], dtype=np.int64)
    starts = np.concatenate([c[1] for c in chunks.values()]).astype(np.int64)
    ends = np.concatenate([c[2] for c in chunks.values()]).astype(np.int64)
    dist = np.concatenate([c[3] for c in chunks.values()]).astype(np.float32)
    maxdiff = np.concatenate([c[4] for c in chunks.values()]).astype(np.int32)
    order = np.lexsort((ends, starts, chrNum))                      # sort_values(by=[chr, binStart, binEnd]) :332
    locationArr = np.stack([chrNum[order], starts[order], ends[order]], axis=1)
    for file in outputDirPath.glob("temp_*.npz"):
        remove(file)
    return locationArr, dist[order], maxdiff[order], chrDict


def _pvals_chunk(x, beta, loc, scale):
    below = np.where(x <= loc)[0]
    above = np.where(x > loc)[0]
    pvals = np.zeros(len(x))
    pvals[below] = 2 * st.gennorm.cdf(x[below], beta, loc=loc, scale=scale)
    pvals[above] = 2 * (1 - st.gennorm.cdf(x[above], beta, loc=loc, scale=scale))
    return pvals


def calculatePVals(distanceArrReal, beta, loc, scale, threads=0):
    """Two-sided p-value of every distance under the fitted gennorm (reference :496-517).  The reference's expression,
    evaluated on slices of the array in threads (scipy's special functions release the GIL; elementwise, so the values are
    the single-threaded ones): 15 M distances took a third of this stage."""
    x = np.asarray(distanceArrReal)
    n = len(x)
    if threads <= 0:
        try:
            import os
            threads = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            threads = cpu_count()
        threads = min(threads, 32)
    step = max(1 << 18, -(-n // max(threads, 1)))
    if n <= step:
        return _pvals_chunk(x, beta, loc, scale)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(lambda lo: _pvals_chunk(x[lo:lo + step], beta, loc, scale), range(0, n, step)))
    return np.concatenate(parts)


def writeMetrics(locationArr, chrDict, maxDiffArr, nameArr, distanceArrReal, outputDirPath, fileTag, pvalBool, pvals=(),
                 mhPvals=()):
    """pairwiseMetrics_{tag}.txt.gz: chr, start, end, state, |distance| %.5f, sign [, p %.5e, BH p %.5e]
    (reference :520-573), through the native multi-threaded writer (same decompressed bytes)."""
    outputDirPath.mkdir(parents=True, exist_ok=True)
    numbers = sorted(chrDict)
    index = np.zeros(max(numbers) + 1 if numbers else 1, dtype=np.int32)
    index[numbers] = np.arange(len(numbers), dtype=np.int32)
    _io.write_metrics(outputDirPath / "pairwiseMetrics_{}.txt.gz".format(fileTag), [chrDict[n] for n in numbers],
                      index[locationArr[:, 0]] if len(locationArr) else np.zeros(0, dtype=np.int32), locationArr[:, 1],
                      locationArr[:, 2], list(nameArr), maxDiffArr, distanceArrReal, pvals if pvalBool else None,
                      mhPvals if pvalBool else None)


def _stars_p(p):
    return "***" if p <= 0.01 else ("**" if p <= 0.05 else ("*" if p <= 0.1 else "."))


def _stars_z(z):
    return "***" if z >= 3 else ("**" if z >= 2 else ("*" if z >= 1 else "."))


def createSignificantLociTxt(filePath, locationArr, chrDict, distanceArr, maxDiffArr, nameArr, pvals, mhPvals):
    """Every bin with BH p <= 0.1 (reference :576-643); the three floats go through float32 like the reference's frame."""
    with gzip.open(filePath, "wt") as out:
        for i in np.where(mhPvals <= 0.1)[0]:
            score = float(np.float32(distanceArr[i]))
            p, mh = float(np.float32(pvals[i])), float(np.float32(mhPvals[i]))
            out.write("{}\t{}\t{}\t{}\t{:.5f}\t{}\t{:.5e}\t{:.5e}\t{}\n".format(
                chrDict[locationArr[i, 0]], locationArr[i, 1], locationArr[i, 2], nameArr[int(maxDiffArr[i]) - 1], abs(score),
                findSign(score), p, mh, _stars_p(mh)))


def _regions(locationArr, distanceArr, roiWidth):
    """Top-100 maxmean windows over |distance| and, per window, the index of its most different bin
    (reference :682-689 / :755-762 with helpers.generateROIIndicesArr :277-296)."""
    W = int(roiWidth)
    absd = np.abs(distanceArr)
    chrom, w_start, w_end, _, centre = maxMean(locationArr[:, 0], locationArr[:, 1], locationArr[:, 2], absd, W, 100)
    maxIndices = np.zeros(len(centre), dtype=np.int64)
    for k, c in enumerate(centre):
        lo = int(c) - W // 2
        hi = int(c) + W // 2 + (1 if W % 2 else 0)
        maxIndices[k] = lo + int(np.argmax(absd[lo:hi]))
    return chrom, w_start, w_end, maxIndices


def createROITxt(filePath, locationArr, chrDict, distanceArr, maxDiffArr, nameArr, pvals, mhPvals, roiWidth):
    """Regions whose most different bin is significant, best first, cut at the first non-significant one
    (reference :646-723)."""
    with open(filePath, "w") as out:
        chrom, w_start, w_end, maxIndices = _regions(locationArr, distanceArr, roiWidth)
        nonSig = np.where(mhPvals[maxIndices] > 0.1)[0]
        n = int(nonSig.min()) if len(nonSig) else len(maxIndices)
        for k in range(n):
            i = maxIndices[k]
            score = float(np.float32(distanceArr[i]))
            p, mh = float(np.float32(pvals[i])), float(np.float32(mhPvals[i]))
            out.write("{}\t{}\t{}\t{}\t{:.5f}\t{}\t{:.5e}\t{:.5e}\t{}\n".format(
                chrDict[int(chrom[k])], int(w_start[k]), int(w_end[k]), nameArr[int(maxDiffArr[i]) - 1], abs(score),
                findSign(score), p, mh, _stars_p(mh)))


def createROINoSignificance(filePath, locationArr, chrDict, distanceArr, maxDiffArr, nameArr, zScores, roiWidth):
    """Top-100 regions with the z-score of their most different bin (reference :726-778)."""
    with open(filePath, "w") as out:
        chrom, w_start, w_end, maxIndices = _regions(locationArr, distanceArr, roiWidth)
        for k in range(len(maxIndices)):
            i = maxIndices[k]
            score = float(np.float32(distanceArr[i]))
            z = float(np.float32(zScores[i]))
            out.write("{}\t{}\t{}\t{}\t{:.5f}\t{}\t{:.5f}\t{}\n".format(
                chrDict[int(chrom[k])], int(w_start[k]), int(w_end[k]), nameArr[int(maxDiffArr[i]) - 1], abs(score),
                findSign(score), z, _stars_z(z)))


def _stage(verbose, label):
    if verbose: print(label + "...", flush=True)
    else: print("    " + label + "\t", end="", flush=True)
    return time()


def _done(verbose, t0):
    import os
    if verbose:
        print("    Time:", time() - t0, flush=True)
    elif os.environ.get("EPILOGOS_TIMING"):
        print("\t[Done] %.2f s" % (time() - t0), flush=True)
    else:
        print("\t[Done]", flush=True)


def _fitParams(distanceArrNull, quiescenceArr, numProcesses, numTrials, samplingSize):
    """Median-likelihood gennorm fit of the non-quiescent null distances (reference :196-221)."""
    data = distanceArrNull[np.where(np.invert(quiescenceArr.astype(bool)))[0]]
    if numProcesses > 1 and numTrials > 1:
        # the reference pickles the whole array into every one of the numTrials tasks (:206-208, 43 MB x 101 at genome scale);
        # forked workers already have it
        global _FIT_DATA
        import multiprocessing
        if multiprocessing.get_start_method(allow_none=True) in (None, "fork"):
            _FIT_DATA = data
            try:
                with closing(multiprocessing.get_context("fork").Pool(numProcesses)) as pool:
                    results = pool.map(_fitOnSharedData, [samplingSize] * numTrials)
                pool.join()
            finally:
                _FIT_DATA = None
        else:
            with closing(Pool(numProcesses)) as pool:
                results = pool.starmap(fitOnSubSample, zip(repeat(data, numTrials), repeat(samplingSize, numTrials)))
            pool.join()
    else:
        results = [fitOnSubSample(data, samplingSize) for _ in range(numTrials)]
    nnlf = np.array([r[1] for r in results], dtype=np.float64)
    return tuple(results[np.argsort(nnlf, kind="stable")[int((numTrials - 1) / 2)]][0])


def _finish(params, locationArr, distanceArrReal, maxDiffArr, chrDict, stateInfo, outputDirPath, fileTag, pvalBool, roiWidth,
            expFreqPath, verbose):
    """Everything of main() after the inputs are in memory (reference :72-169 without the figures)."""
    stateNameList = getStateNames(stateInfo)
    roiPath = outputDirPath / "regionsOfInterest_{}.txt".format(fileTag)
    if pvalBool:
        beta, loc, scale = params[0], params[-2], params[-1]
        t0 = _stage(verbose, "Calculating p-vals")
        pvals = calculatePVals(distanceArrReal, beta, loc, scale)
        _done(verbose, t0)
        t0 = _stage(verbose, "Benjamini-Hochberg procedure")
        mhPvals = benjaminiHochberg(pvals)
        _done(verbose, t0)
        t0 = _stage(verbose, "Writing metrics")
        writeMetrics(locationArr, chrDict, maxDiffArr, stateNameList, distanceArrReal, outputDirPath, fileTag, True,
                     pvals=pvals, mhPvals=mhPvals)
        _done(verbose, t0)
        t0 = _stage(verbose, "Regions of interest txt")
        createROITxt(roiPath, locationArr, chrDict, distanceArrReal, maxDiffArr, stateNameList, pvals, mhPvals, roiWidth)
        _done(verbose, t0)
        t0 = _stage(verbose, "Significant loci txt")
        createSignificantLociTxt(outputDirPath / "significantLoci_{}.txt.gz".format(fileTag), locationArr, chrDict,
                                 distanceArrReal, maxDiffArr, stateNameList, pvals, mhPvals)
        _done(verbose, t0)
    else:
        t0 = _stage(verbose, "Z-Scores")
        zScores = np.abs(st.zscore(distanceArrReal))
        _done(verbose, t0)
        t0 = _stage(verbose, "Writing metrics")
        writeMetrics(locationArr, chrDict, maxDiffArr, stateNameList, distanceArrReal, outputDirPath, fileTag, False)
        _done(verbose, t0)
        t0 = _stage(verbose, "Regions of interest txt")
        createROINoSignificance(roiPath, locationArr, chrDict, distanceArrReal, maxDiffArr, stateNameList, zScores, roiWidth)
        _done(verbose, t0)
    remove(Path(expFreqPath))


def _is_location_sorted(chrNum, starts, ends):
    """True when rows are already in non-decreasing (chromosome, start, end) order -- then the stable lexsort is the identity."""
    if len(chrNum) < 2:
        return True
    c0, c1, s0, s1 = chrNum[:-1], chrNum[1:], starts[:-1], starts[1:]
    up = (c1 > c0) | ((c1 == c0) & ((s1 > s0) | ((s1 == s0) & (ends[1:] >= ends[:-1]))))
    return bool(up.all())


def mainFromArrays(results, stateInfo, outputDir, fileTag, numProcesses, pvalBool, numTrials, samplingSize, expFreqPath,
                   roiWidth, verbose):
    """main() on the arrays driver.run_paired_groups hands back ({stem: dict(chrName, locations, nullDistances,
    quiescenceArr, distances, maxDiff)}) instead of the temp_*.npz / pairwiseDelta text round trip."""
    outputDirPath = Path(outputDir)
    if numProcesses == 0:
        numProcesses = cpu_count()
    byChr = {v["chrName"]: v for v in results.values()}
    chrOrder = orderChromosomes(list(byChr))
    cat = lambda key: np.concatenate([byChr[c][key] for c in chrOrder])
    params = None
    if pvalBool:
        t0 = _stage(verbose, "Fitting distances")
        params = _fitParams(cat("nullDistances"), cat("quiescenceArr"), numProcesses, numTrials, samplingSize)
        _done(verbose, t0)
    cols = [byChr[c]["locations"].columns() for c in chrOrder]
    chrNum = np.concatenate([np.full(len(c[1]), i + 1, dtype=np.int64) for i, c in enumerate(cols)])
    starts, ends = np.concatenate([c[1] for c in cols]), np.concatenate([c[2] for c in cols])
    chrDict = {i + 1: c for i, c in enumerate(chrOrder)}
    dist, maxdiff = cat("distances"), cat("maxDiff")
    if _is_location_sorted(chrNum, starts, ends):
        # the usual case -- every file lists its bins in genomic order --: the stable three-key sort of 15 M rows (reference :332,
        # half of this stage's time at genome scale) is the identity
        locationArr = np.stack([chrNum, starts, ends], axis=1)
    else:
        order = np.lexsort((ends, starts, chrNum))
        locationArr = np.stack([chrNum[order], starts[order], ends[order]], axis=1)
        dist, maxdiff = dist[order], maxdiff[order]
    _finish(params, locationArr, dist, maxdiff, chrDict, stateInfo, outputDirPath, fileTag, pvalBool, roiWidth, expFreqPath, verbose)


def main(group1Name, group2Name, stateInfo, outputDir, fileTag, numProcesses, pvalBool, diagnosticBool, numTrials,
         samplingSize, expFreqPath, roiWidth, verbose, backend=None):
    tTotal = time()
    outputDirPath = Path(outputDir)
    numStates = getNumStates(stateInfo)
    if numProcesses == 0:
        numProcesses = cpu_count()
    params = None
    if pvalBool:
        t0 = _stage(verbose, "Fitting distances")
        params, _, _ = fitDistances(outputDirPath, numProcesses, numTrials, samplingSize)
        _done(verbose, t0)
    t0 = _stage(verbose, "Reading in files")
    locationArr, distanceArrReal, maxDiffArr, chrDict = readInData(outputDirPath, numStates, backend)
    _done(verbose, t0)
    _finish(params, locationArr, distanceArrReal, maxDiffArr, chrDict, stateInfo, outputDirPath, fileTag, pvalBool, roiWidth,
            expFreqPath, verbose)
    if verbose: print("Total Time:", time() - tTotal, flush=True)


if __name__ == "__main__":
    main(argv[1], argv[2], argv[3], argv[4], argv[5], int(argv[6]), strToBool(argv[7]), strToBool(argv[8]), int(argv[9]),
         int(argv[10]), argv[11], int(argv[12]), strToBool(argv[13]))

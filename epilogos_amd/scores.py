"""STEP 3 -- per-file scores.  Same call signature, argv form and on-disk artefacts as the reference's
epilogos/scores.py (main :14-56, calculateScores :116-169, calculateScoresPairwise :172-256, writeScores :509-536):
  single: scores_{tag}_{stem}.txt.gz and temp_scores_{tag}_{stem}.npz {chrName, scoreArr float32, locationArr object}
  paired: pairwiseDelta_{tag}_{stem}.txt.gz, temp_nullDistances_{tag}_{stem}.npz, temp_quiescence_{tag}_{stem}.npz
All arithmetic runs on the GPU through the C ABI."""
from pathlib import Path
from sys import argv
from time import time

import numpy as np

from . import _io
from . import backend as _backend
from .helpers import countRows, fileStem, readStates, readTable, strToBool

def klScoreND(obs, exp):
    """obs * log2(obs / exp) with the reference's masked-array semantics (scores.py:539-550): the quotient is masked where
    exp == 0 or it is not finite and filled with 0; the logarithm is masked where its argument is <= 0 and filled with 0."""
    obs = np.asarray(obs)
    quotient = np.ma.divide(obs, exp).filled(0)
    return obs * np.ma.log2(quotient).filled(0)


def s1ScoreTable(expFreqArr, numCols):
    """The S1 score of every possible count: T[c, s] = klScoreND(c / numCols, expFreqArr[s]) for c = 0 .. numCols
    (scores.py:317 with rowObsS1's `count / dataArr.shape[1]`, :327-344) -- float64 [numCols + 1, S] and its float32 store.
    Evaluated with numpy on the host, i.e. by the very expression and the very log2 the reference runs per bin, so the
    GPU score pass that gathers from it (epg_score_s1_from_binhist_table) returns the reference's float32 values bit for
    bit; 834 x 18 entries for the EpiMap matrix."""
    obs = (np.arange(numCols + 1, dtype=np.int64) / numCols)[:, None]            # int64 / int: numpy's float64 true division
    t64 = np.ascontiguousarray(klScoreND(obs, np.asarray(expFreqArr, dtype=np.float32)[None, :]), dtype=np.float64)
    return t64, t64.astype(np.float32)


NULL_SEED = None   # paired nulls are unseeded in the reference (helpers.py:183); set an int for reproducible runs


def main(file1, file2, numStates, saliency, outputDir, expFreqPath, fileTag, numProcesses, quiescentState, groupSize,
         verbose):
    _io.set_state_limit(numStates)
    if verbose: tTotal = time()
    file1Path, file2Path, outputDirPath = Path(file1), Path(file2), Path(outputDir)
    filename = fileStem(file1Path)
    if not verbose: print("    {}\t".format(filename), end="", flush=True)
    totalRows = countRows(file1Path)
    if str(file2) == "null":
        calculateScores(saliency, file1Path, totalRows, numStates, outputDirPath, expFreqPath, fileTag, filename, verbose)
    else:
        calculateScoresPairwise(saliency, file1Path, file2Path, totalRows, numStates, outputDirPath, expFreqPath,
                                fileTag, filename, quiescentState, groupSize, verbose)
    print("Total Time:", time() - tTotal, flush=True) if verbose else print("\t[Done]", flush=True)


def calculateScores(saliency, file1Path, totalRows, numStates, outputDirPath, expFreqPath, fileTag, filename, verbose):
    if saliency not in (1, 2, 3):
        raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")
    expFreqArr = np.load(expFreqPath, allow_pickle=False)
    dataArr, locations = readTable(file1Path, (0, totalRows))
    scoreArr = _backend.get().scores(dataArr, numStates, saliency, expFreqArr)
    writeScores(scoreArr, outputDirPath / "scores_{}_{}.txt.gz".format(fileTag, filename), locations)
    locationArr = locations.to_object_array()
    chrName = locationArr[0, 0]
    np.savez_compressed(outputDirPath / "temp_scores_{}_{}.npz".format(fileTag, filename), chrName=np.array([chrName]),
                        scoreArr=scoreArr, locationArr=locationArr)


def calculateScoresPairwise(saliency, file1Path, file2Path, totalRows, numStates, outputDirPath, expFreqPath, fileTag,
                            filename, quiescentState, groupSize, verbose):
    if saliency not in (1, 2):
        raise ValueError("Please ensure that saliency metric is either 1 or 2 for Pairwise Epilogos")
    be = _backend.get()
    expFreqArr = np.load(expFreqPath, allow_pickle=False)
    file1Arr, locations = readTable(file1Path, (0, totalRows))
    file2Arr = readStates(file1Path=file2Path, rowsToCalc=(0, totalRows), verbose=verbose)
    n1, n2 = file1Arr.shape[1], file2Arr.shape[1]
    perms1, perms2 = n1 * (n1 - 1), n2 * (n2 - 1)
    score1 = be.scores(file1Arr, numStates, saliency, expFreqArr, perms=perms1)
    score2 = be.scores(file2Arr, numStates, saliency, expFreqArr, perms=perms2)
    seed = NULL_SEED if NULL_SEED is not None else int(np.random.SeedSequence().generate_state(1)[0])
    null1, null2 = be.null_scores(file1Arr, file2Arr, numStates, saliency, expFreqArr, groupSize, seed)
    realDiffArr, _ = be.pair_finish(score1, score2)
    _, nullDistancesArr = be.pair_finish(null1, null2)
    quiescenceArr = be.quiescent(file1Arr, file2Arr, quiescentState)

    writeScores(realDiffArr, outputDirPath / "pairwiseDelta_{}_{}.txt.gz".format(fileTag, filename), locations)
    chrName = locations.slice(0, 1).to_object_array()[0, 0]
    np.savez_compressed(outputDirPath / "temp_nullDistances_{}_{}.npz".format(fileTag, filename),
                        chrName=np.array([chrName]), nullDistances=nullDistancesArr)
    np.savez_compressed(outputDirPath / "temp_quiescence_{}_{}.npz".format(fileTag, filename),
                        chrName=np.array([chrName]), quiescenceArr=quiescenceArr)


def writeScores(dataArr, outputTxtPath, locationArr, gzip_level=None, threads=0):
    """gzip text file, one line per bin: 'chr\\tstart\\tend\\t' + '%.5f' values (reference scores.py:509-536).
    Native writer: exact '%.5f' of the float32 values, one gzip member per 32768 rows compressed in
    parallel; the decompressed bytes equal the reference's.  locationArr: a _io.Locations (verbatim input columns) or
    the reference's [rows, 3] object array."""
    if not isinstance(locationArr, _io.Locations):
        locationArr = _io.Locations.from_object_array(locationArr)
    # gzip_level None: EPILOGOS_GZIP_LEVEL, by default 0 = the library's own compressor (csrc/epg_deflate.h), ~5x the speed of
    # zlib level 6 for files 4-12 % larger -- writing the text is most of a whole-genome run; 1..9 select zlib (the reference's
    # gzip.open default is level 9: 3 % smaller than 6, five times its time)
    _io.write_scores(outputTxtPath, locationArr, np.asarray(dataArr, dtype=np.float32), threads=threads, gzip_level=gzip_level)


if __name__ == "__main__":
    main(argv[1], argv[2], int(argv[3]), int(argv[4]), argv[5], argv[6], argv[7], int(argv[8]), int(argv[9]),
         int(argv[10]), strToBool(argv[11]))

"""STEP 1 -- per-file background state counts.  Same call signature, argv form and on-disk artefact as the
reference's epilogos/expected.py (main :11-45, storeExpArray :207-223): writes
OUT/temp_exp_freq_{fileTag}_{stem}.npy holding int64[S] (S1), int64[S,S] (S2) or int32[N,N,S,S] (S3).
The counting itself runs on the GPU (epg_bin_hist / epg_hist_s2_from_binhist / epg_hist_s3)."""
from pathlib import Path
from sys import argv
from time import time

import numpy as np

from . import _io
from . import backend as _backend
from .helpers import countRows, fileStem, readStates, strToBool


def main(file1, file2, numStates, saliency, outputDir, fileTag, numProcesses, verbose):
    """file2 == "null" for single-group runs; numProcesses is accepted for CLI compatibility and ignored (the
    row-range fan-out of the reference's Pool is the GPU's grid)."""
    _io.set_state_limit(numStates)
    if verbose: tTotal = time()
    file1Path, file2Path, outputDirPath = Path(file1), Path(file2), Path(outputDir)
    filename = fileStem(file1Path)
    if not verbose: print("    {}\t".format(filename), end="", flush=True)
    if saliency not in (1, 2, 3):
        raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")
    if saliency == 3 and str(file2Path) != "null":
        raise ValueError("Saliency 3 is not supported for paired epilogos")

    totalRows = countRows(file1Path)
    dataArr = readStates(file1Path=file1Path, file2Path=file2Path, rowsToCalc=(0, totalRows), verbose=verbose)
    be = _backend.get()
    expFreqArr = be.expected_counts(dataArr, numStates, saliency)
    be.check_counts(expFreqArr, dataArr.shape[0], dataArr.shape[1], saliency)
    storeExpArray(expFreqArr, outputDirPath, fileTag, filename)
    print("Total Time:", time() - tTotal, flush=True) if verbose else print("\t[Done]", flush=True)


def storeExpArray(expFreqArr, outputDirPath, fileTag, filename):
    np.save(Path(outputDirPath) / "temp_exp_freq_{}_{}.npy".format(fileTag, filename), expFreqArr, allow_pickle=False)


if __name__ == "__main__":
    main(argv[1], argv[2], int(argv[3]), int(argv[4]), argv[5], argv[6], int(argv[7]), strToBool(argv[8]))

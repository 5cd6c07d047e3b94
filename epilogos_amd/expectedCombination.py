"""STEP 2 -- combine the per-file counts into the background frequencies.  Same signature and artefact as the
reference's epilogos/expectedCombination.py (main :9-46): sums OUT/temp_exp_freq_{fileTag}_*.npy, removes the
temporaries (all tags, like the reference: quirk Q3), writes storedExpInput = float32(counts / sum(counts))."""
from os import remove
from pathlib import Path
from sys import argv
from time import time

import numpy as np

from . import backend as _backend
from .helpers import strToBool


def main(outputDirectory, storedExpInput, fileTag, verbose):
    if verbose: tTotal = time()
    outputDirPath, storedExpPath = Path(outputDirectory), Path(storedExpInput)
    expFreqArr = None
    for file in sorted(outputDirPath.glob("temp_exp_freq_{}_*.npy".format(fileTag))):
        part = np.load(file, allow_pickle=False)
        expFreqArr = part if expFreqArr is None else expFreqArr + part
    if expFreqArr is None:
        raise FileNotFoundError("no temp_exp_freq_{}_*.npy in {}".format(fileTag, outputDirPath))
    for file in outputDirPath.glob("temp_exp_freq_*.npy"):
        remove(file)
    expFreqArr = _backend.get().normalise(expFreqArr)
    np.save(storedExpPath, expFreqArr, allow_pickle=False)
    print("Total Time:", time() - tTotal) if verbose else print("    [Done]")


if __name__ == "__main__":
    main(argv[1], argv[2], argv[3], strToBool(argv[4]))

"""Host helpers of the scoring path -- same names and argument meaning as the reference's epilogos/helpers.py
(getNumStates :9-17, strToBool :47-60, countRows :80-99, splitRows :102-120, readStates :123-194), minus the pieces that
belong to ROI/plotting.  States come back as int8 (0-based) instead of the reference's int64: that is the layout the
GPU kernels stream."""
import gzip
from pathlib import Path

import numpy as np
import pandas as pd


def getNumStates(stateFile):
    """Number of states = data rows of the state-metadata TSV (reference helpers.py:9-17)."""
    return pd.read_table(Path(stateFile), header=0, sep="\t").shape[0]


def strToBool(string):
    """'True'/'False' -> bool, anything else raises (reference helpers.py:47-60)."""
    if string == "True":
        return True
    if string == "False":
        return False
    raise ValueError("Invalid boolean string")


def countRows(dataFilePath):
    """Number of newline characters in a (gz) file (reference helpers.py:80-99).  Like the reference, a final line
    without a trailing newline is not counted (quirk Q6)."""
    dataFilePath = Path(dataFilePath)
    opener = gzip.open if dataFilePath.name.endswith("gz") else open
    total = 0
    with opener(dataFilePath, "rb") as f:
        while True:
            block = f.read(1 << 20)
            if not block:
                break
            total += block.count(b"\n")
    return total


def splitRows(totalRows, numProcesses):
    """Contiguous ranges (i*R//P, (i+1)*R//P) (reference helpers.py:102-120).  Also the GPU bin-range partition."""
    return [(i * totalRows // numProcesses, (i + 1) * totalRows // numProcesses) for i in range(numProcesses)]


def _read_int8(path, rowsToCalc):
    path = Path(path)
    ncols = pd.read_table(path, nrows=1, header=None, sep="\t").shape[1]
    nrows = rowsToCalc[1] - rowsToCalc[0]
    if nrows <= 0:
        return np.zeros((0, ncols - 3), dtype=np.int8)
    df = pd.read_table(path, usecols=range(3, ncols), skiprows=rowsToCalc[0], nrows=nrows, header=None, sep="\t",
                       dtype=np.int16)
    arr = df.to_numpy(dtype=np.int16) - 1          # file states are 1-based (reference helpers.py:154-155)
    if arr.size and (arr.min() < -128 or arr.max() > 127):
        raise ValueError("state value out of int8 range in {}".format(path))
    return arr.astype(np.int8)


def readStates(file1Path=Path("null"), file2Path=Path("null"), rowsToCalc=(0, 0), expBool=True, verbose=True,
               groupSize=-1, rng=None):
    """Reference helpers.py:123-194.  Single: int8 [rows, N].  Paired + expBool: column concatenation [A|B].
    Paired scores: (A, B, shuffledA, shuffledB) where the shuffle is a per-row uniform permutation of [A|B]
    (argsort of i.i.d. uniforms, helpers.py:183-184) split at N_A, or into two halves of `groupSize`.
    The host shuffle here exists for API parity and tests; the engine shuffles on device (epg_null_hist)."""
    file1Arr = _read_int8(file1Path, rowsToCalc)
    if str(file2Path) == "null":
        return file1Arr
    file2Arr = _read_int8(file2Path, rowsToCalc)
    combinedArr = np.concatenate((file1Arr, file2Arr), axis=1)
    if expBool:
        return combinedArr
    rand = (rng.random(combinedArr.shape) if rng is not None else np.random.rand(*combinedArr.shape))
    shuffled = np.take_along_axis(combinedArr, np.argsort(rand, axis=1), axis=1)
    if groupSize == -1:
        return file1Arr, file2Arr, shuffled[:, :file1Arr.shape[1]], shuffled[:, file1Arr.shape[1]:]
    return file1Arr, file2Arr, shuffled[:, :groupSize], shuffled[:, groupSize:2 * groupSize]


def readLocations(filePath, rowsToCalc=None):
    """First three columns (chromosome, start, end), echoed verbatim into the outputs (reference scores.py:161)."""
    kw = {}
    if rowsToCalc is not None:
        kw = dict(skiprows=rowsToCalc[0], nrows=rowsToCalc[1] - rowsToCalc[0])
    return pd.read_table(Path(filePath), header=None, sep="\t", usecols=[0, 1, 2], **kw).to_numpy()


def fileStem(path):
    """Output stem = text before the first '.' of the file name (reference expected.py:31, scores.py:38; quirk Q2)."""
    return Path(path).name.split(".")[0]

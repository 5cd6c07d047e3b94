"""Host helpers of the scoring path -- same names and argument meaning as the reference's epilogos/helpers.py
(getNumStates :9-17, strToBool :47-60, countRows :80-99, splitRows :102-120, readStates :123-194), minus the pieces that
belong to ROI/plotting.  States come back as int8 (0-based) instead of the reference's int64: that is the layout the
GPU kernels stream."""
import os
from pathlib import Path

import numpy as np

from . import _io


def getNumStates(stateFile):
    """Number of states = data rows of the state-metadata TSV (reference helpers.py:9-17: pandas' read_table with a header
    row, whose tokenizer skips a line that is empty or holds nothing but SPACES -- a line of tabs, the separator, is a row of
    empty fields -- counted here without importing pandas, a third of a second the command line spends before it has read a
    byte otherwise)."""
    with open(Path(stateFile), "r", newline=None) as fh:
        rows = [l for l in fh.read().splitlines() if l.strip(" ") != ""]
    if not rows:
        import pandas as pd
        return pd.read_table(Path(stateFile), header=0, sep="\t").shape[0]      # (raises pandas' EmptyDataError like the reference)
    return len(rows) - 1


def strToBool(string):
    """'True'/'False' -> bool, anything else raises (reference helpers.py:47-60)."""
    if string == "True":
        return True
    if string == "False":
        return False
    raise ValueError("Invalid boolean string")


def countRows(dataFilePath):
    """Number of newline characters in a (gz) file (reference helpers.py:80-99).  Like the reference, a final line
    without a trailing newline is not counted (quirk Q6).  Native (zlib) reader."""
    return _io.count_rows(Path(dataFilePath))


def splitRows(totalRows, numProcesses):
    """Contiguous ranges (i*R//P, (i+1)*R//P) (reference helpers.py:102-120).  Also the GPU bin-range partition."""
    return [(i * totalRows // numProcesses, (i + 1) * totalRows // numProcesses) for i in range(numProcesses)]


def _cache_paths(path):
    """Binary cache of a parsed matrix file: EPILOGOS_CACHE_DIR/<key>.{states,locblob,locoff}.npy, keyed
    by the file's absolute path, size and modification time so that a changed input is parsed again."""
    root = os.environ.get("EPILOGOS_CACHE_DIR")
    if not root:
        return None
    import hashlib
    st = os.stat(path)
    # (a cache parsed for a model of up to 31 states holds -1 for larger values: the wide models get their own)
    wide = "|wide" if _io.state_limit() > 31 else ""
    key = hashlib.sha1("{}|{}|{}{}".format(os.path.abspath(path), st.st_size, st.st_mtime_ns, wide).encode()).hexdigest()[:20]
    base = Path(root) / "{}_{}".format(Path(path).name.split(".")[0], key)
    return [Path(str(base) + ext) for ext in (".states.npy", ".locblob.npy", ".locoff.npy", ".range.npy")]


def _readTablePandas(path, rowsToCalc, alloc, with_range):
    """A matrix file as the REFERENCE reads it -- pandas.read_table(header=None, sep="\\t"), helpers.py:152-155 for the states and
    scores.py:161 for the first three columns -- for a file the native parser refuses.  pandas is more lenient than a strict
    reading of the format (README.md:127-134): it skips blank lines, ignores blanks around a number, takes "+1" and "1.0" (a
    float column cast to int).  A drop-in must read what the reference reads, so such a file goes through pandas itself (slow:
    ~2 us per value) with a warning; what pandas cannot turn into integers raises here as it does there.  Locations come back
    as the reference would print them ("{}\\t{}\\t{}" of pandas' values, scores.py:526-531)."""
    import pandas as pd
    df = pd.read_table(Path(path), header=None, sep="\t")
    body = df.iloc[:, 3:]
    if body.isna().to_numpy().any():
        # a row with fewer fields than the first: pandas pads it with NaN and the reference's to_numpy(dtype=int) turns that into
        # an arbitrary integer (a numpy RuntimeWarning, no error) -- a wrong result; here it is an error
        raise ValueError("{}: a row has fewer state columns than the first row".format(path))
    vals = body.to_numpy(dtype=int)                                        # (the reference subtracts 1 here: helpers.py:155)
    R = vals.shape[0]
    lo, hi = (0, R) if rowsToCalc is None else (max(rowsToCalc[0], 0), min(rowsToCalc[1], R))
    hi = max(hi, lo)
    limit = _io.state_limit()
    part = vals[lo:hi]
    st = np.where((part >= 1) & (part <= limit), part - 1, -1).astype(np.int8)     # outside the model: "not a state", like the native parser
    if alloc is None:
        out = st
    else:
        out = alloc(hi - lo, st.shape[1])
        out[:, :st.shape[1]] = st
        out[:, st.shape[1]:] = -1
    loc = _io.Locations.from_object_array(df.iloc[lo:hi, :3].to_numpy())
    if with_range:
        return out, loc, ((int(vals.min()), int(vals.max())) if vals.size else (0, 0))
    return out, loc


def _readTableNative(path, rowsToCalc, threads, alloc, with_range):
    """_io.read_table; a file with a line the native parser calls malformed is read again the way the reference reads it."""
    try:
        return _io.read_table(Path(path), rowsToCalc, threads=threads, alloc=alloc, with_range=with_range)
    except _io.EpilogosIOError as e:
        if "malformed line" not in str(e):
            raise
        print("epilogos_amd: {} -- reading this file through pandas like the reference does (slow)".format(e), flush=True)
        return _readTablePandas(path, rowsToCalc, alloc, with_range)


def readTable(path, rowsToCalc=None, alloc=None, with_range=False, threads=0):
    """Rows [lo, hi) of a matrix file through the native multi-threaded parser: int8 0-based states
    [rows, N] and the rows' first three columns as written (a _io.Locations); a file the strict native parser refuses but pandas
    reads (blank lines, blanks around numbers, "+1", "1.0") is read through pandas like the reference's (_readTablePandas).
    With EPILOGOS_CACHE_DIR set (the command
    line's --cache-dir) the parsed file is kept as an int8 [R, N] .npy plus the coordinate side-car and later runs on the
    same input memory-map it instead of inflating and parsing ~1.7 KB of text per bin again.
    alloc(R, N) -> int8 [R, width >= N] supplies the destination (the driver's pinned, row-padded staging; columns >= N
    are set to -1); with_range also returns the (lowest, highest) state value of the WHOLE file as written (1-based);
    threads = native threads for this one file (0 = all cores; the driver, which reads many files at once, gives each its share)."""
    cache = _cache_paths(path)
    if cache is None:
        return _readTableNative(path, rowsToCalc, threads, alloc, with_range)
    if not all(c.exists() for c in cache):
        # first run on this input: parse, serve the caller from the arrays just parsed, and write the cache files BEHIND the
        # caller's back (a whole genome is 12.9 GB of them; written before the part was handed on they were a third of a cold
        # run).  The writer threads are ordinary (non-daemon) threads: the interpreter waits for them at exit.
        states, loc, rng = _readTableNative(path, None, threads, None, True)
        blob, offsets, rng_arr = loc.blob, loc.offsets, np.array(rng, dtype=np.int64)
        _save_cache_async(cache, (states, blob, offsets, rng_arr))
    else:
        states = np.load(cache[0], mmap_mode="r")
        blob, offsets = np.load(cache[1], mmap_mode="r"), np.load(cache[2], mmap_mode="r")
        rng_arr = np.load(cache[3]) if with_range else None
    loc = _io.Locations(blob, offsets)
    lo, hi = (0, states.shape[0]) if rowsToCalc is None else (max(rowsToCalc[0], 0), min(rowsToCalc[1], states.shape[0]))
    hi = max(hi, lo)
    part = loc.slice(lo, hi)
    if alloc is None:
        out = np.ascontiguousarray(states[lo:hi])
    else:
        out = alloc(hi - lo, states.shape[1])
        out[:, :states.shape[1]] = states[lo:hi]
        out[:, states.shape[1]:] = -1
    ploc = _io.Locations(np.ascontiguousarray(part.blob), np.ascontiguousarray(part.offsets))
    if with_range:
        return out, ploc, (int(rng_arr[0]), int(rng_arr[1]))
    return out, ploc


_cache_writers = {}                  # cache key (first file of the set) -> writer thread
_cache_lock = __import__("threading").Lock()


def _save_cache_async(cache, arrays):
    """Writes the cache files of one input behind the caller's back.  One writer per cache key and process (paired mode can
    read the same path for both groups, the driver parses in many threads); temp names carry pid AND a per-writer token, and
    reach their final name by os.replace, so concurrent ranks or threads can only ever publish complete files."""
    import threading
    import uuid
    key = str(cache[0])
    token = "%d.%s" % (os.getpid(), uuid.uuid4().hex[:12])

    def work():
        tmps = []
        try:
            cache[0].parent.mkdir(parents=True, exist_ok=True)
            for c, arr in zip(cache, arrays):
                tmp = Path(str(c) + ".tmp%s.npy" % token)
                tmps.append(tmp)
                np.save(tmp, arr, allow_pickle=False)
                os.replace(tmp, c)
        except OSError as e:                                               # a cache that cannot be written is not an error of the run
            print("epilogos_amd: could not write the input cache {}: {}".format(cache[0], e), flush=True)
        finally:
            for tmp in tmps:                                               # whatever did not reach its final name
                try:
                    tmp.unlink()
                except OSError:
                    pass

    with _cache_lock:
        th = _cache_writers.get(key)
        if th is not None and th.is_alive():
            return                                                         # this input's cache is being written already
        th = threading.Thread(target=work, name="epilogos-cache-writer")
        _cache_writers[key] = th
        th.start()


def flushCacheWrites():
    """Wait for the cache files of this process's first-time reads: end of main, the error exits of the command line (which
    leave through os._exit and would otherwise kill the writers half-way), tests, callers that read the cache back at once."""
    while True:
        with _cache_lock:
            if not _cache_writers:
                return
            _key, th = _cache_writers.popitem()
        th.join()


def _read_int8(path, rowsToCalc):
    if rowsToCalc[1] - rowsToCalc[0] <= 0:
        import pandas as pd
        ncols = pd.read_table(Path(path), nrows=1, header=None, sep="\t").shape[1]
        return np.zeros((0, ncols - 3), dtype=np.int8)
    return readTable(path, rowsToCalc)[0]


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
    """First three columns (chromosome, start, end) as an object array like the reference's locationArr
    (scores.py:161)."""
    return readTable(filePath, rowsToCalc)[1].to_object_array()


def fileStem(path):
    """Output stem = text before the first '.' of the file name (reference expected.py:31, scores.py:38; quirk Q2)."""
    return Path(path).name.split(".")[0]

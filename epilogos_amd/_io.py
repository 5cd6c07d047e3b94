"""ctypes binding of include/epilogos_io.h (native TSV parser + score writer, host side)."""
import ctypes as C
import io
import os

import numpy as np

from . import build as _build

_lib = None
# int8_t* alloc(int64_t rows, int32_t cols, int64_t* ldx, void* user): include/epilogos_io.h epgio_alloc_fn
_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.c_void_p)


class EpilogosIOError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    # EPILOGOS_IO_LIB: another build of the same source (tools/asan_io.sh points it at the AddressSanitizer build)
    path = _build.IO_LIB_PATH if not os.environ.get("EPILOGOS_IO_LIB") else __import__("pathlib").Path(os.environ["EPILOGOS_IO_LIB"])
    if not path.exists():
        raise EpilogosIOError("%s is missing: build it with `python -m epilogos_amd.build`" % path)
    lib = C.CDLL(str(path))
    p, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
    lib.epgio_last_error.restype = C.c_char_p
    lib.epgio_count_rows.restype = i64
    lib.epgio_count_rows.argtypes = [C.c_char_p]
    lib.epgio_open_table.restype = p
    lib.epgio_open_table.argtypes = [C.c_char_p, i64, i64, i32]
    lib.epgio_open_table_ex.restype = p
    lib.epgio_open_table_ex.argtypes = [C.c_char_p, i64, i64, i32, i32]
    lib.epgio_open_table_into.restype = p
    lib.epgio_open_table_into.argtypes = [C.c_char_p, i64, i64, i32, i32, _ALLOC_FN, p]
    lib.epgio_table_rows.restype = i64
    lib.epgio_table_rows.argtypes = [p]
    lib.epgio_table_cols.restype = i32
    lib.epgio_table_cols.argtypes = [p]
    lib.epgio_table_state_range.restype = C.c_int
    lib.epgio_table_state_range.argtypes = [p, C.POINTER(i32), C.POINTER(i32)]
    lib.epgio_table_copy_states.restype = C.c_int
    lib.epgio_table_copy_states.argtypes = [p, p, i64]
    lib.epgio_table_locations.restype = p
    lib.epgio_table_locations.argtypes = [p, C.POINTER(p)]
    lib.epgio_close_table.restype = None
    lib.epgio_close_table.argtypes = [p]
    lib.epgio_parse_locations.restype = C.c_int
    lib.epgio_parse_locations.argtypes = [p, p, i64, p, p, C.POINTER(i32), i32]
    lib.epgio_write_scores.restype = C.c_int
    lib.epgio_write_scores.argtypes = [C.c_char_p, p, p, p, i64, i32, i32, i32]
    lib.epgio_write_states.restype = C.c_int
    lib.epgio_write_states.argtypes = [C.c_char_p, C.c_char_p, i64, i64, p, i64, i32, i64, i32, i32]
    lib.epgio_write_metrics.restype = C.c_int
    lib.epgio_write_metrics.argtypes = [C.c_char_p, p, p, p, p, p, p, p, p, p, p, p, i64, i32, i32]
    lib.epgio_format_f5.restype = i64
    lib.epgio_format_f5.argtypes = [p, i64, C.c_char, p, i64]
    lib.epgio_gzip_fast.restype = i64
    lib.epgio_gzip_fast.argtypes = [p, i64, p, i64]
    lib.epgio_row_sums_f32.restype = C.c_int
    lib.epgio_row_sums_f32.argtypes = [p, i64, C.c_int32, i64, p, C.c_int32]
    lib.epgio_rolling_max_f64.restype = C.c_int
    lib.epgio_rolling_max_f64.argtypes = [p, i64, C.c_int32, p, C.c_int32]
    lib.epgio_inflate_mem.restype = i64
    lib.epgio_inflate_mem.argtypes = [p, i64, p, i64, i32]
    lib.epgio_release_buffers.restype = None
    lib.epgio_release_buffers.argtypes = [i32]
    lib.epgio_set_reader_plan.restype = None
    lib.epgio_set_reader_plan.argtypes = [i32]
    lib.epgio_default_threads.restype = i32
    lib.epgio_default_threads.argtypes = []
    lib.epgio_set_host_threads.restype = None
    lib.epgio_set_host_threads.argtypes = [i32]
    lib.epgio_thread_census.restype = None
    lib.epgio_thread_census.argtypes = [C.POINTER(i32), C.POINTER(i32), i32]
    _lib = lib
    return lib


def _err():
    return load().epgio_last_error().decode(errors="replace")


def node_cores():
    """Cores the processes of this job may run on at once: the scheduler affinity capped by the cgroup CPU quota (a GPU box
    shows a container 256 hardware threads and lets it run 16 at a time)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


_budget_told = None


def host_budget():
    """Native threads THIS rank may keep busy: the node's cores (node_cores), capped by the reference's -c/--num-cores when the
    user gave one (EPILOGOS_NUM_CORES; run.py:36,148 -- there it sizes the worker pool), divided by the ranks that share the
    node (LOCAL_WORLD_SIZE, set by torch.distributed.run), at least 1.  Exported as EPILOGOS_HOST_THREADS, which is what the
    native library uses wherever its `threads` argument is 0 (parser, writers, STEP 4 helpers)."""
    n = node_cores()
    try:
        cap = int(os.environ.get("EPILOGOS_NUM_CORES", "0"))
    except ValueError:
        cap = 0
    if cap > 0:
        n = min(n, cap)
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    b = max(1, n // local_world)
    # the native library gets the value through a setter, not through the environment: its reader threads must never read
    # `environ` while this thread (or torch, imported next to the early readers) writes it.  The variable is still exported, once
    # per value, for child processes (the launcher's ranks) and the thread census.
    global _budget_told
    if _budget_told != b:
        _budget_told = b
        load().epgio_set_host_threads(b)
        if os.environ.get("EPILOGOS_HOST_THREADS") != str(b):
            os.environ["EPILOGOS_HOST_THREADS"] = str(b)
    return b


def inflate_mem(blob, own=True, cap=None):
    """Inflate a gzip stream held in memory with the library's own inflate (own=True), its parallel BGZF reader (own=2: declines
    whatever is not blocked gzip all the way) or zlib's (own=False); -> bytes, or None when the stream is declined / corrupt.  For
    tests and the fuzz harness."""
    blob = bytes(blob)
    cap = cap if cap is not None else max(1 << 16, 64 * len(blob))
    out = C.create_string_buffer(cap)
    n = load().epgio_inflate_mem(blob, len(blob), out, cap, 2 if own == 2 and own is not True else (1 if own else 0))
    return None if n < 0 else out.raw[:n]


def release_buffers(background=True):
    load().epgio_release_buffers(1 if background else 0)


def set_reader_plan(n):
    load().epgio_set_reader_plan(int(n))


def thread_census(reset=False):
    """(library threads runnable right now, their peak since the last reset)."""
    live, peak = C.c_int32(0), C.c_int32(0)
    load().epgio_thread_census(C.byref(live), C.byref(peak), 1 if reset else 0)
    return int(live.value), int(peak.value)


def log_thread_census(tag=""):
    """EPILOGOS_THREAD_LOG=<file>: one line per rank -- pid, peak of runnable library threads, this rank's budget, the node's
    cores, LOCAL_WORLD_SIZE -- appended when a driver run ends (the multi-rank tests and profiles/ add the ranks up)."""
    log = os.environ.get("EPILOGOS_THREAD_LOG")
    if log:
        _live, peak = thread_census()
        with open(log, "a") as f:
            f.write("%d\t%d\t%s\t%d\t%s\t%s\n" % (os.getpid(), peak, os.environ.get("EPILOGOS_HOST_THREADS", "-"), node_cores(),
                                                  os.environ.get("LOCAL_WORLD_SIZE", "1"), tag))


_state_limit = 31


def set_state_limit(numStates):
    """The state model the process works with.  Up to 31 states the parser stores file values outside 1..31 as "not a state"
    (the fast kernels decode five bits of a byte); a larger model keeps 1..127 and runs the wide kernels (epg_wide.hip)."""
    global _state_limit
    _state_limit = 127 if int(numStates) > 31 else 31


def state_limit():
    return _state_limit


def _log_io(op, path, lo=0, hi=-1):
    """EPILOGOS_IO_LOG=<file>: one line per pass over an input file (every call below inflates the whole file); the
    multi-rank tests count them."""
    log = os.environ.get("EPILOGOS_IO_LOG")
    if log:
        with open(log, "a") as f:
            f.write("%d\t%s\t%s\t%d\t%d\n" % (os.getpid(), op, os.path.abspath(str(path)), lo, hi))


def count_rows(path):
    _log_io("count", path)
    n = load().epgio_count_rows(str(path).encode())
    if n < 0:
        raise EpilogosIOError(_err())
    return int(n)


class Locations:
    """First three columns of every row as the file wrote them: a byte blob ("chr\\tstart\\tend\\n" per row) + offsets."""

    def __init__(self, blob, offsets):
        self.blob = blob            # np.uint8 [total]
        self.offsets = offsets      # np.int64 [rows + 1]

    def __len__(self):
        return len(self.offsets) - 1

    def slice(self, lo, hi):
        off = self.offsets[lo:hi + 1]
        return Locations(self.blob[off[0]:off[-1]], (off - off[0]).astype(np.int64))

    def to_object_array(self):
        """[rows, 3] object array like the reference's locationArr (pandas-parsed: scores.py:161)."""
        import pandas as pd
        if len(self) == 0:
            return np.empty((0, 3), dtype=object)
        return pd.read_table(io.BytesIO(self.blob.tobytes()), header=None, sep="\t").to_numpy()

    def columns(self):
        """(chromosome object array, start int64, end int64) of every row.  Native parse when the coordinates are plain
        integers and every row names the same chromosome (one file = one chromosome); pandas otherwise."""
        import pandas as pd
        R = len(self)
        if R == 0:
            return np.empty(0, dtype=object), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        blob = np.ascontiguousarray(self.blob)
        off = np.ascontiguousarray(self.offsets, dtype=np.int64)
        start, end, same = np.empty(R, dtype=np.int64), np.empty(R, dtype=np.int64), C.c_int32(0)
        rc = load().epgio_parse_locations(blob.ctypes.data, off.ctypes.data, R, start.ctypes.data, end.ctypes.data, C.byref(same), 0)
        if rc == 0 and same.value:
            first = blob[off[0]:off[1]].tobytes()
            chrom = np.empty(R, dtype=object)
            chrom[:] = first[:first.index(b"\t")].decode()
            return chrom, start, end
        df = pd.read_table(io.BytesIO(self.blob.tobytes()), header=None, sep="\t")
        return df.iloc[:, 0].to_numpy(dtype=object), df.iloc[:, 1].to_numpy(dtype=np.int64), df.iloc[:, 2].to_numpy(dtype=np.int64)

    def start_end(self):
        _, s, e = self.columns()
        return s, e

    @staticmethod
    def from_object_array(loc):
        rows = ["{}\t{}\t{}\n".format(r[0], r[1], r[2]).encode() for r in loc]
        off = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum([len(r) for r in rows], out=off[1:])
        return Locations(np.frombuffer(b"".join(rows), dtype=np.uint8).copy(), off)


def read_table(path, rows=None, threads=0, ldx=None, alloc=None, with_range=False):
    """Parse rows [lo, hi) of a TSV(.gz) matrix file.  Returns (int8 states [R, ldx or N], Locations) and, with_range, the
    (lowest, highest) state value as written in the file.  alloc(R, N) -> int8 array [R, width >= N] lets the caller own
    the destination (e.g. a pinned, row-padded staging buffer); columns >= N are filled with -1."""
    lib = load()
    lo, hi = (0, -1) if rows is None else rows
    _log_io("read", path, lo, hi)
    dest, failure = [], []

    def check(states, R, N):
        if states.dtype != np.int8 or states.ndim != 2 or states.shape[0] != R or states.shape[1] < N \
                or (R and states.strides != (states.shape[1], 1)):
            raise ValueError("alloc must return a C-contiguous int8 [R, width >= N] array")
        return states

    if alloc is not None:
        # the native reader calls back once it knows the shape (after the inflate and the line count) and parses every row
        # straight into the caller's array -- a page-locked staging buffer in the driver: no intermediate matrix, no copy
        def native_alloc(R, N, ldx_out, _user):
            try:
                states = check(alloc(int(R), int(N)), int(R), int(N))
                dest.append(states)
                ldx_out[0] = states.shape[1]
                return states.ctypes.data or 1          # (an empty array may have no address: any non-NULL value will do)
            except BaseException as e:                  # must not propagate through the C frames
                failure.append(e)
                return None
        cb = _ALLOC_FN(native_alloc)
        h = lib.epgio_open_table_into(str(path).encode(), lo, hi, threads, _state_limit, cb, None)
    else:
        h = lib.epgio_open_table_ex(str(path).encode(), lo, hi, threads, _state_limit)
    if failure:
        if h:
            lib.epgio_close_table(h)
        raise failure[0]
    if not h:
        raise EpilogosIOError(_err())
    try:
        R, N = lib.epgio_table_rows(h), lib.epgio_table_cols(h)
        if alloc is not None:                           # (an empty file returns before the reader asks for a destination)
            states = dest[0] if dest else check(alloc(R, N), R, N)
        else:
            width = N if ldx is None else ldx
            states = np.empty((R, width), dtype=np.int8)
            if lib.epgio_table_copy_states(h, states.ctypes.data, width) != 0:
                raise EpilogosIOError(_err())
        slo, shi = C.c_int32(0), C.c_int32(0)
        lib.epgio_table_state_range(h, C.byref(slo), C.byref(shi))
        offp = C.c_void_p()
        locp = lib.epgio_table_locations(h, C.byref(offp))
        off = np.ctypeslib.as_array(C.cast(offp, C.POINTER(C.c_int64)), shape=(R + 1,)).copy()
        blob = (np.ctypeslib.as_array(C.cast(locp, C.POINTER(C.c_uint8)), shape=(int(off[-1]),)).copy()
                if R else np.zeros(0, dtype=np.uint8))
    finally:
        lib.epgio_close_table(h)
    if with_range:
        return states, Locations(blob, off), (int(slo.value), int(shi.value))
    return states, Locations(blob, off)


def default_gzip_level():
    """0 = the library's own fast compressor (csrc/epg_deflate.h); EPILOGOS_GZIP_LEVEL=1..9 selects zlib."""
    try:
        v = int(os.environ.get("EPILOGOS_GZIP_LEVEL", "0"))
    except ValueError:
        v = 0
    return v if 0 <= v <= 9 else 0


def write_scores(path, locations, scores, threads=0, gzip_level=None):
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    R, S = scores.shape
    if len(locations) != R:
        raise ValueError("locations and scores disagree on the number of rows")
    blob = np.ascontiguousarray(locations.blob)
    off = np.ascontiguousarray(locations.offsets, dtype=np.int64)
    rc = load().epgio_write_scores(str(path).encode(), blob.ctypes.data, off.ctypes.data, scores.ctypes.data, R, S,
                                   threads, default_gzip_level() if gzip_level is None else gzip_level)
    if rc != 0:
        raise EpilogosIOError(_err())


def write_states(path, chrom, states, start0=0, step=200, n_cols=None, threads=0, gzip_level=1):
    """Write an int8 [R, width] matrix of 0-based states in the reference's input format (1-based text, .gz by name)."""
    states = np.asarray(states)
    if states.dtype != np.int8 or states.ndim != 2 or (states.shape[0] and states.strides != (states.shape[1], 1)):
        raise ValueError("states must be a C-contiguous int8 [R, width] array")
    R, ldx = states.shape
    N = ldx if n_cols is None else n_cols
    rc = load().epgio_write_states(str(path).encode(), str(chrom).encode(), start0, step, states.ctypes.data, R, N, ldx,
                                   threads, gzip_level)
    if rc != 0:
        raise EpilogosIOError(_err())


def _string_table(strings):
    enc = [str(x).encode() for x in strings]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(e) for e in enc], out=off[1:])
    return np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy(), off


def write_metrics(path, chrom_names, chrom_idx, start, end, state_names, maxdiff, dist, pvals=None, mh=None, threads=0,
                  gzip_level=None):
    """pairwiseMetrics text (see epilogos_io.h): chrom_names[chrom_idx[r]], start, end, state_names[maxdiff[r] - 1],
    |dist| %.5f, sign [, p %.5e, adjusted p %.5e]."""
    R = len(dist)
    cb, co = _string_table(chrom_names)
    nb, no = _string_table(state_names)
    ci = np.ascontiguousarray(chrom_idx, dtype=np.int32)
    st_, en = np.ascontiguousarray(start, dtype=np.int64), np.ascontiguousarray(end, dtype=np.int64)
    md = np.ascontiguousarray(maxdiff, dtype=np.int32)
    di = np.ascontiguousarray(dist, dtype=np.float32)
    if R and (ci.min() < 0 or ci.max() >= len(chrom_names) or md.min() < 1 or md.max() > len(state_names)):
        raise ValueError("chromosome or state index out of range")
    pv = None if pvals is None else np.ascontiguousarray(pvals, dtype=np.float64)
    mv = None if mh is None else np.ascontiguousarray(mh, dtype=np.float64)
    rc = load().epgio_write_metrics(str(path).encode(), cb.ctypes.data, co.ctypes.data, ci.ctypes.data, st_.ctypes.data, en.ctypes.data,
                                    nb.ctypes.data, no.ctypes.data, md.ctypes.data, di.ctypes.data,
                                    None if pv is None else pv.ctypes.data, None if mv is None else mv.ctypes.data, R, threads,
                                    default_gzip_level() if gzip_level is None else gzip_level)
    if rc != 0:
        raise EpilogosIOError(_err())


def row_sums(scores, threads=0):
    """scores.sum(axis=1) of a float32 [R, S] matrix with numpy's exact rounding (eight accumulators, tree, remainder)."""
    a = np.asarray(scores)
    if a.dtype != np.float32 or a.ndim != 2 or a.shape[1] > 128 or a.shape[1] < 1 or a.strides[1] != 4 or a.strides[0] % 4:
        return np.asarray(scores).sum(axis=1)
    out = np.empty(a.shape[0], dtype=np.float32)
    if load().epgio_row_sums_f32(a.ctypes.data, a.shape[0], a.shape[1], a.strides[0] // 4, out.ctypes.data, threads) != 0:
        raise EpilogosIOError(_err())
    return out


def rolling_max(x, window, threads=0):
    """pandas' Series(x).rolling(window, center=True).max() for a float64 vector without NaNs."""
    v = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(v.size, dtype=np.float64)
    if load().epgio_rolling_max_f64(v.ctypes.data, v.size, int(window), out.ctypes.data, threads) != 0:
        raise EpilogosIOError(_err())
    return out


def gzip_fast(data):
    """One gzip member of `data` from the library's own DEFLATE compressor (what the writers use at gzip_level 0)."""
    src = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.empty(src.size + src.size // 8 + 1100, dtype=np.uint8)
    n = load().epgio_gzip_fast(src.ctypes.data if src.size else None, src.size, out.ctypes.data, out.size)
    if n < 0:
        raise EpilogosIOError(_err())
    return out[:n].tobytes()


def format_f5(values, sep="\t"):
    v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    buf = np.empty(48 * v.size + 1, dtype=np.uint8)
    n = load().epgio_format_f5(v.ctypes.data, v.size, sep.encode(), buf.ctypes.data, buf.size)
    if n < 0:
        raise EpilogosIOError(_err())
    return buf[:n].tobytes()

"""ctypes binding of include/epilogos_amd.h.  No fallbacks: a missing library or symbol raises."""
import ctypes as C
import re
from pathlib import Path

from . import build as _build

EPG_OK = 0
ERR_NAMES = {-1: "EPG_ERR_INVALID_ARG", -2: "EPG_ERR_UNSUPPORTED", -3: "EPG_ERR_HIP", -4: "EPG_ERR_WORKSPACE"}

HEADER = Path(__file__).resolve().parents[1] / "include" / "epilogos_amd.h"


class EpilogosHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "EPG_ERR"), code, msg))
        self.code = code


_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_u64 = C.c_uint64

# name -> (restype, argtypes); must list every prototype of the header (tests/test_abi_symbols.py checks it)
PROTOTYPES = {
    "epg_version": (C.c_int, []),
    "epg_last_error": (C.c_char_p, []),
    "epg_device_cus": (C.c_int, []),
    "epg_bin_hist": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p, _p]),
    "epg_bin_hist_s2": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p, _p, _p]),
    "epg_bin_hist_parts": (C.c_int, [_i32, _p, _p, _p, _p, _i32, _p, _p, _p]),
    "epg_hist_s1": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p]),
    "epg_hist_s2": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p, _i64, _p]),
    "epg_hist_s2_from_binhist": (C.c_int, [_p, _i64, _i32, _p, _p]),
    "epg_hist_s2_from_binhist_pair": (C.c_int, [_p, _p, _i64, _i32, _p, _p]),
    "epg_combine_score_s1": (C.c_int, [_p, _i32, _p, _i64, _i32, _i32, _p, _p, _p, _p, _i64, _p]),
    "epg_hist_s3": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p, _i64, _p]),
    "epg_normalise_i64": (C.c_int, [_p, _i64, _p, _p, _i64, _p]),
    "epg_normalise_i32": (C.c_int, [_p, _i64, _p, _p, _i64, _p]),
    "epg_ws_bytes": (_i64, [_i32, _i64, _i32, _i32]),
    "epg_score_s1": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "epg_score_s1_from_binhist": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p, _p, _i64, _p]),
    "epg_score_s1_from_binhist_table": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p, _p, _p]),
    "epg_pair_scores_s1_from_binhist": (C.c_int, [_p, _p, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "epg_pair_scores_s1_parts": (C.c_int, [_i32, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "epg_score_s2": (C.c_int, [_p, _i64, _i32, _i64, _i32, _i64, _p, _p, _p, _p, _i64, _p]),
    "epg_score_s2_from_binhist": (C.c_int, [_p, _i64, _i32, _i32, _i64, _p, _p, _p, _p, _i64, _p]),
    "epg_score_s3": (C.c_int, [_p, _i64, _i32, _i64, _i32, _p, _p, _p, _p, _i64, _p]),
    "epg_pair_finish": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p]),
    "epg_pair_metrics": (C.c_int, [_p, _i64, _i32, _i32, _p, _p, _p]),
    "epg_quiescent": (C.c_int, [_p, _i32, _i64, _p, _i32, _i64, _i64, _i32, _p, _p]),
    "epg_null_hist": (C.c_int, [_p, _i32, _i64, _p, _i32, _i64, _i64, _i32, _i32, _i32, _u64, _i64, _p, _p, _p]),
    "epg_quiescent_from_binhist": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "epg_null_hist_from_binhist": (C.c_int, [_p, _p, _i64, _i32, _i32, _i32, _i32, _u64, _i64, _p, _p, _p]),
    "epg_null_hist_from_binhist_parts": (C.c_int, [_i32, _p, _p, _p, _i32, _i32, _i32, _i32, _u64, _p, _p, _p, _p]),
    "epg_pair_count_null_parts": (C.c_int, [_i32, _p, _p, _p, _i32, _i32, _p, _p, _i32, _p, _p, _p, _u64, _p, _p, _p, _p]),
    "epg_test_force": (C.c_int, [_i32, _i32]),
}

_lib = None


def header_symbols():
    """Function names declared in include/epilogos_amd.h."""
    txt = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(epg_[a-z0-9_]+)\s*\(", txt)))


def lib_path():
    """The in-tree library; EPILOGOS_HIP_LIB=<path> loads another build of it (A/B measurements of kernel variants)."""
    import os
    p = os.environ.get("EPILOGOS_HIP_LIB")
    return Path(p) if p else _build.LIB_PATH


def _header_abi_version():
    m = re.search(r"^#define\s+EPG_ABI_VERSION\s+(\d+)", HEADER.read_text(), flags=re.M)
    if m is None:
        raise RuntimeError("EPG_ABI_VERSION not found in %s" % HEADER)
    return int(m.group(1))


ABI_VERSION = _header_abi_version()                         # one source: the header the library is built from


def load():
    """Load libepilogos_hip.so (must have been built: __graft_entry__.build() or python -m epilogos_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; it must be the HIP runtime of this process BEFORE our library resolves
    # its NEEDED libamdhip64.so.7, otherwise two runtimes coexist and our launches see no device.
    import torch  # noqa: F401
    path = lib_path()
    if not path.exists():
        raise EpilogosHipError(-3, "%s is missing: build it with `python -m epilogos_amd.build` "
                                   "(there is no CPU fallback)" % path)
    lib = C.CDLL(str(path))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.epg_version() != ABI_VERSION:                    # (EPG_ABI_VERSION of include/epilogos_amd.h the library was built from)
        raise EpilogosHipError(-3, "%s has ABI version %d, this package speaks %d: rebuild it with `python -m epilogos_amd.build`"
                               % (path, lib.epg_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        raise EpilogosHipError(rc, load().epg_last_error().decode(errors="replace"))
    return rc


def call(name, *args):
    return check(getattr(load(), name)(*args))

"""Build libepilogos_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "_lib"
LIB_PATH = LIB_DIR / "libepilogos_hip.so"
SOURCES = ["epg_abi.hip", "epg_s1.hip", "epg_s2.hip", "epg_s3.hip", "epg_s3_transpose.hip", "epg_s3_gemm.hip", "epg_s3_lanes.hip", "epg_null.hip", "epg_wide.hip"]
HEADERS = [CSRC / "epg_common.h", CSRC / "epg_count.h", ROOT / "include" / "epilogos_amd.h"]
ARCH = "gfx950"


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libepilogos_hip.so")
    return exe


def is_stale():
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + HEADERS
    return any(d.stat().st_mtime > t for d in deps)


IO_LIB_PATH = LIB_DIR / "libepilogos_io.so"
IO_SOURCE = CSRC / "epg_io.cpp"
IO_HEADER = ROOT / "include" / "epilogos_io.h"


def io_is_stale():
    if not IO_LIB_PATH.exists():
        return True
    t = IO_LIB_PATH.stat().st_mtime
    return any(f.stat().st_mtime > t for f in (IO_SOURCE, IO_HEADER, CSRC / "epg_deflate.h", CSRC / "epg_inflate.h", CSRC / "epg_crc32.h"))


def build_io_library(force=False, verbose=False):
    """Host-side native TSV parser / score writer (g++, zlib, pthreads)."""
    if not force and not io_is_stale():
        return IO_LIB_PATH
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found: cannot build libepilogos_io.so")
    tmp = LIB_DIR / (IO_LIB_PATH.name + ".tmp.%d" % os.getpid())
    cmd = [cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + str(ROOT / "include"), str(IO_SOURCE),
           "-lz", "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, IO_LIB_PATH)
    return IO_LIB_PATH


OBJ_DIR = LIB_DIR / "obj"


def _compile_one(src, obj, extra_flags, verbose):
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c",
           "-I" + str(ROOT / "include"), "-I" + str(CSRC), *extra_flags, str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (src.name, res.stdout, res.stderr))


def build_library(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source into one shared library (one object per source, the stale ones in parallel, then one
    link).  Returns the path.  EPILOGOS_BUILD_EXPERIMENTS=1 in the environment builds with -DEPILOGOS_BUILD_EXPERIMENTS: the
    measurement switches of the kernels (EPG_S3_DBG, EPG_S3_SCORE_DBG, EPG_S3_KC, EPG_S3_AHEAD, EPG_S3_MFMA, EPG_PAIR_WAVES) then
    read the environment; the default library never does."""
    if os.environ.get("EPILOGOS_BUILD_EXPERIMENTS") == "1":
        extra_flags = tuple(extra_flags) + ("-DEPILOGOS_BUILD_EXPERIMENTS",)
    if not force and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    flag_tag = OBJ_DIR / "flags.txt"
    flags_now = " ".join(extra_flags)
    same_flags = flag_tag.exists() and flag_tag.read_text() == flags_now
    hdr_t = max(h.stat().st_mtime for h in HEADERS)
    todo, objs = [], []
    for s in SOURCES:
        src, obj = CSRC / s, OBJ_DIR / (s + ".o")
        objs.append(obj)
        if force or not same_flags or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_t):
            todo.append((src, obj))
    jobs = max(1, min(len(todo), int(os.environ.get("EPILOGOS_BUILD_JOBS", os.cpu_count() or 4))))
    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            for f in [pool.submit(_compile_one, src, obj, extra_flags, verbose) for src, obj in todo]:
                f.result()
        flag_tag.write_text(flags_now)
    tmp = LIB_DIR / (LIB_PATH.name + ".tmp.%d" % os.getpid())
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared", *[str(o) for o in objs], "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_io_library(force=True, verbose=True))

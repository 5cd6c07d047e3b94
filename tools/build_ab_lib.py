#!/usr/bin/env python3
"""Build container: another build of libepilogos_hip.so into the git-ignored tools/_ab_libs/ (it travels to the GPU box with the
snapshot), for tools/s1_ab.py / s3_ab.py / EPILOGOS_HIP_LIB.
usage: build_ab_lib.py NAME.so [--experiments] [--rev GITREV] [-DFLAG ...]
  --experiments   -DEPILOGOS_BUILD_EXPERIMENTS: the measurement switches (EPG_S3_OVERLAP, EPG_S3_AUX_PRIO, EPG_S3_SCORE_DBG ...) read the environment
  --rev REV       build the sources of that commit (git worktree under /tmp) instead of the working tree"""
import os, shutil, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
args = sys.argv[1:]
name = args.pop(0)
flags, rev = [], None
while args:
    a = args.pop(0)
    if a == "--experiments":
        flags.append("-DEPILOGOS_BUILD_EXPERIMENTS")
    elif a == "--rev":
        rev = args.pop(0)
    else:
        flags.append(a)
src_root = ROOT
if rev:
    src_root = Path(tempfile.mkdtemp(prefix="epg_ab_"))
    subprocess.run(["git", "-C", str(ROOT), "worktree", "add", "--detach", str(src_root), rev], check=True, capture_output=True)
try:
    csrc = src_root / "epilogos_amd" / "csrc"
    sources = sorted(csrc.glob("*.hip"))
    out = ROOT / "tools" / "_ab_libs"
    out.mkdir(parents=True, exist_ok=True)
    objdir = Path(tempfile.mkdtemp(prefix="epg_ab_obj_"))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

    def one(s):
        o = objdir / (s.name + ".o")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-I" + str(src_root / "include"), "-I" + str(csrc),
                        *flags, str(s), "-o", str(o)], check=True)
        return o
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as pool:
        objs = list(pool.map(one, sources))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *map(str, objs), "-o", str(out / name)], check=True)
    print(out / name)
finally:
    if rev:
        subprocess.run(["git", "-C", str(ROOT), "worktree", "remove", "--force", str(src_root)], capture_output=True)

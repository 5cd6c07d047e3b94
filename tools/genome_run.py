#!/usr/bin/env python3
"""Whole-genome-scale run of the command line (reference flow: run.py:190-279 processes every file of the input directory):
24 synthetic "chromosome" .txt.gz files with the hg19 bin counts (200 bp bins, 15 478 375 in total) x 833 biosamples x 18
states, written in the reference's input format by the native writer, then
    python -m epilogos_amd.run -l -i <dir> -j <metadata> -o <out> -s 1 --cache-dir <cache>
cold (text parse, fills the binary cache) and warm (memory-maps the cache), on one GPU.  Prints the phase table
(EPILOGOS_TIMING), wall time and peak host RSS of each run, checks every output's line count and spot-checks scores against
the engine.  usage: genome_run.py [--scale 1.0] [--dir /dev/shm/epg_genome] [--saliency 1] [--keep]"""
import argparse
import gzip
import json
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

HG19 = [("chr1", 249250621), ("chr2", 243199373), ("chr3", 198022430), ("chr4", 191154276), ("chr5", 180915260),
        ("chr6", 171115067), ("chr7", 159138663), ("chr8", 146364022), ("chr9", 141213431), ("chr10", 135534747),
        ("chr11", 135006516), ("chr12", 133851895), ("chr13", 115169878), ("chr14", 107349540), ("chr15", 102531392),
        ("chr16", 90354753), ("chr17", 81195210), ("chr18", 78077248), ("chr19", 59128983), ("chr20", 63025520),
        ("chr21", 48129895), ("chr22", 51304566), ("chrX", 155270560), ("chrY", 59373566)]

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0, help="fraction of every chromosome's bins (quick trials)")
ap.add_argument("--dir", default=None)
ap.add_argument("--saliency", type=int, default=1)
ap.add_argument("--biosamples", type=int, default=833)
ap.add_argument("--keep", action="store_true")
ap.add_argument("--skip-warm", action="store_true")
ap.add_argument("--also", default="", help="comma list of further saliencies to run from the warm cache, e.g. 2,3")
ap.add_argument("--paired", action="store_true", help="also write the 379 + 342 column split as two groups and run -m paired (S1)")
ap.add_argument("--ranks", type=int, default=0, help="also: a cold run WITHOUT the cache on this many ranks (`--gpus N`, gloo transport, the "
                                                   "ranks share the GPU) next to a cold one-rank run without the cache; thread census per rank")
ap.add_argument("--bgzf", action="store_true", help="write the inputs as BGZF (blocked gzip): the reader inflates one file's blocks in parallel")
ap.add_argument("--only-ranks", action="store_true", help="stop after the --ranks comparison")
ap.add_argument("--timeline", action="store_true", help="per-part reader timeline and per-file reader phases of the cold runs (stderr of the CLI)")
ap.add_argument("--pvals", action="store_true", help="with --paired: one more warm run with -n (null-distribution fit, p-values, BH)")
a = ap.parse_args()
N, S = a.biosamples, 18
base = Path(a.dir or (Path("/dev/shm") if Path("/dev/shm").is_dir() else Path(os.environ.get("TMPDIR", "/tmp"))) / "epg_genome")
shutil.rmtree(base, ignore_errors=True)
ind, cache = base / "in", base / "cache"
ind.mkdir(parents=True)
gA, gB = base / "male", base / "female"
if a.paired:
    gA.mkdir(); gB.mkdir()
meta = base / "metadata.tsv"
meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\tS%d\n" % (i, i + 1, i + 1) for i in range(S)))

if a.bgzf:
    os.environ["EPILOGOS_BGZF"] = "1"            # for the input writer below only
# ---- inputs: states drawn on the GPU (bench.generate_shard: chr1 frequencies, fixed global chunk seeds), text by the native writer
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import _io, engine  # noqa: E402
engine.require_gpu()
t0 = time.time()
rows, bin0, text_bytes = {}, 0, 0
for name, bp in HG19:
    R = max(int((bp // 200 + (1 if bp % 200 else 0)) * a.scale), 1)
    X = torch.empty((R, N), dtype=torch.int8, device="cuda")
    bench.generate_shard(torch, X, N, S, bin0)
    x = X.cpu().numpy()
    del X
    path = ind / ("matrix_%s.txt.gz" % name)
    _io.write_states(path, name, x, gzip_level=1)
    if a.paired:                                     # BASELINE config 5 shape: 379 + 342 biosamples
        _io.write_states(gA / path.name, name, np.ascontiguousarray(x[:, :379]), gzip_level=1)
        _io.write_states(gB / path.name, name, np.ascontiguousarray(x[:, 379:721]), gzip_level=1)
    rows[name] = R
    bin0 += R
    text_bytes += path.stat().st_size
os.environ.pop("EPILOGOS_BGZF", None)
total = bin0
print("inputs: %d files, %d bins x %d biosamples, %.2f GB of .txt.gz written in %.1f s" % (len(HG19), total, N, text_bytes / 1e9, time.time() - t0), flush=True)
torch.cuda.empty_cache()

WRAP = ("import resource, subprocess, sys, json, time; t = time.time(); r = subprocess.run(sys.argv[1:]); "
        "print('@@' + json.dumps({'rc': r.returncode, 'wall_s': time.time() - t, "
        "'peak_rss_gb': resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1048576.0})); sys.exit(r.returncode)")


def run(label, out, saliency=None, paired=False, extra=(), use_cache=True, gpus=1, timeline=False):
    src = ["-m", "paired", "-a", str(gA), "-b", str(gB), "--null-seed", "7", *extra] if paired else ["-i", str(ind)]
    cmd = [sys.executable, "-c", WRAP, sys.executable, "-m", "epilogos_amd.run", "-l", *src, "-j", str(meta), "-o", str(out),
           "-s", str(saliency or a.saliency)] + (["--cache-dir", str(cache)] if use_cache else []) + (["--gpus", str(gpus)] if gpus > 1 else [])
    env = dict(os.environ, EPILOGOS_TIMING=os.environ.get("EPILOGOS_TIMING", "2" if timeline else "1"))
    if timeline:
        env["EPGIO_TIMING"] = "1"
    tlog = base / ("threads_%d.log" % len(list(base.glob("threads_*.log"))))
    env["EPILOGOS_THREAD_LOG"] = str(tlog)
    if gpus > 1:
        env["EPILOGOS_DIST_BACKEND"] = "gloo"                   # several ranks on the one GPU of this box
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
    r = subprocess.run(cmd, cwd=str(ROOT), capture_output=True, text=True, env=env)
    if tlog.exists():
        census = [l.split("\t") for l in tlog.read_text().splitlines()]
        print("== %s: native threads -- per rank peak runnable / budget: %s; sum of peaks %d of %s node cores" % (
            label, ", ".join("%s/%s" % (c[1], c[2]) for c in census), sum(int(c[1]) for c in census), census[0][3] if census else "?"))
    info = [json.loads(l[2:]) for l in r.stdout.splitlines() if l.startswith("@@")]
    print("== %s" % label)
    print("".join(l + "\n" for l in r.stdout.splitlines() if "[timing]" in l or ("[Done]" in l and " s" in l)), end="")
    if timeline or os.environ.get("EPGIO_TIMING") or os.environ.get("EPILOGOS_TIMING") == "2":   # reader phases / part timeline (stderr)
        print("".join(l + "\n" for l in r.stderr.splitlines() if "[epgio]" in l or "[part" in l or "[pool]" in l), end="")
    if r.returncode != 0 or not info:
        print(r.stdout[-3000:], r.stderr[-5000:])
        raise SystemExit("%s failed" % label)
    i = info[0]
    print("%s: %.1f s wall for %d bins = %.3f Mbins/s end to end (process start, parse, upload, kernels, %%.5f text + gzip, STEP 4); "
          "peak host RSS %.1f GB" % (label, i["wall_s"], total, total / i["wall_s"] / 1e6, i["peak_rss_gb"]), flush=True)
    return i


out1, out2 = base / "out_cold", base / "out_warm"
if a.ranks:
    c1 = run("cold, no cache, 1 rank", base / "out_cold_nc1", use_cache=False, timeline=a.timeline)
    cN = run("cold, no cache, %d ranks on the one GPU (gloo)" % a.ranks, base / "out_cold_ncN", use_cache=False, gpus=a.ranks, timeline=a.timeline)
    for name in ("chr1", "chr21", "chrY"):
        fn = "scores_in_s%d_matrix_%s.txt.gz" % (a.saliency, name)
        assert gzip.open(base / "out_cold_nc1" / fn, "rb").read() == gzip.open(base / "out_cold_ncN" / fn, "rb").read(), "1 rank vs %d ranks: %s" % (a.ranks, name)
    print("1 rank vs %d ranks: chr1 / chr21 / chrY outputs identical; wall %.1f s vs %.1f s" % (a.ranks, c1["wall_s"], cN["wall_s"]), flush=True)
    shutil.rmtree(base / "out_cold_nc1", ignore_errors=True)
    shutil.rmtree(base / "out_cold_ncN", ignore_errors=True)
    if a.only_ranks:
        shutil.rmtree(base, ignore_errors=True)
        raise SystemExit(0)
cold = run("cold (inflate + parse text, fills the cache)", out1, timeline=a.timeline)
warm = None if a.skip_warm else run("warm (memory-mapped int8 cache)", out2)

for sal in [int(v) for v in a.also.split(",") if v]:
    o = base / ("out_s%d" % sal)
    run("saliency %d from the warm cache" % sal, o, saliency=sal)
    for name, _ in HG19:
        with gzip.open(o / ("scores_in_s%d_matrix_%s.txt.gz" % (sal, name)), "rb") as fh:
            n = sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b""))
        assert n == rows[name], (sal, name, n, rows[name])

if a.paired:
    o = base / "out_paired"
    run("paired S1, 379 + 342 biosamples, cold", o, saliency=1, paired=True)
    run("paired S1, 379 + 342 biosamples, warm cache", base / "out_paired2", saliency=1, paired=True)
    if a.pvals:
        run("paired S1 with -n (gennorm fit of the null distances, p-values, Benjamini-Hochberg), warm cache", base / "out_paired3",
            saliency=1, paired=True, extra=("-n", "-c", "16"))
    for name, _ in HG19:
        with gzip.open(o / ("pairwiseDelta_male_female_s1_matrix_%s.txt.gz" % name), "rb") as fh:
            n = sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b""))
        assert n == rows[name], ("paired", name, n, rows[name])
    with gzip.open(o / "pairwiseMetrics_male_female_s1.txt.gz", "rb") as fh:
        n = sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b""))
    assert n == total, ("pairwiseMetrics", n, total)
    a_ = gzip.open(o / "pairwiseDelta_male_female_s1_matrix_chr21.txt.gz", "rb").read()
    b_ = gzip.open(base / "out_paired2" / "pairwiseDelta_male_female_s1_matrix_chr21.txt.gz", "rb").read()
    assert a_ == b_, "paired cold and warm runs differ"
    print("paired checks OK: 24 pairwiseDelta files and pairwiseMetrics with the right line counts", flush=True)

# ---- checks: every chromosome's output has its number of lines; chr21 equals the engine's scores of the cached matrix
tag = "in_s%d" % a.saliency
for name, _ in HG19:
    p = out1 / ("scores_%s_matrix_%s.txt.gz" % (tag, name))
    with gzip.open(p, "rb") as fh:
        n = sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b""))
    assert n == rows[name], (name, n, rows[name])
if warm is not None:
    for name in ("chr21", "chrY"):
        a_ = gzip.open(out1 / ("scores_%s_matrix_%s.txt.gz" % (tag, name)), "rb").read()
        b_ = gzip.open(out2 / ("scores_%s_matrix_%s.txt.gz" % (tag, name)), "rb").read()
        assert a_ == b_, "cold and warm runs differ on " + name
if a.saliency == 1:
    counts = torch.zeros(S, dtype=torch.int64, device="cuda")
    H21 = None
    for name, _ in HG19:
        st = [np.load(p, mmap_mode="r") for p in cache.glob("matrix_%s_*.states.npy" % name)]
        x = [m for m in st if m.shape[1] == N][0]                  # the paired runs cache their 379- and 342-column files too
        Xd = engine.states_to_device(np.ascontiguousarray(x))
        H, _ = engine.bin_hist(Xd, N, S, counts=counts)
        if name == "chr21":
            H21 = H
        del Xd
    q = engine.normalise(counts)
    o32, _ = engine.score_s1_from_binhist(H21, N, S, q)
    with gzip.open(out1 / ("scores_%s_matrix_chr21.txt.gz" % tag), "rt") as fh:
        got = np.loadtxt(fh, usecols=range(3, 3 + S), max_rows=4000, dtype=np.float64)
    k = min(4000, rows["chr21"])
    assert np.allclose(got[:k], o32.cpu().numpy()[:k], atol=1.01e-5), "CLI text differs from the engine's scores"
    print("checks OK: %d output files with the right line counts; chr21 text == engine scores; exp_freq from all 24 files" % len(HG19))
if not a.keep:
    shutil.rmtree(base, ignore_errors=True)

#!/usr/bin/env python3
"""End-to-end run of the command line at the reference's real width: synthetic per-chromosome .txt.gz matrices
(833 biosamples, 18 states, chr1 state frequencies) -> `python -m epilogos_amd.run` -> the reference's output files,
with wall-clock per mode.  Also a paired run on the 379 + 342 split.  usage: e2e_bench.py [--bins 200000] [--files 3]"""
import argparse
import gzip
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
FREQS = np.array([.00570, .00293, .00430, .00212, .03260, .10464, .00154, .00057, .01001, .00416, .01554, .00618,
                  .02498, .00262, .00140, .01412, .05563, .71097])
ap = argparse.ArgumentParser()
ap.add_argument("--bins", type=int, default=200000)
ap.add_argument("--files", type=int, default=3)
ap.add_argument("--s3-bins", type=int, default=20000)
ap.add_argument("--keep", action="store_true")
ap.add_argument("--only", default="", help="comma list of labels to run: s1,s2,s3,paired,pairedn")
a = ap.parse_args()
N, S = 833, 18
tmp = Path(os.environ.get("TMPDIR", "/tmp")) / "epg_e2e"
shutil.rmtree(tmp, ignore_errors=True)
(tmp / "all").mkdir(parents=True)
(tmp / "male").mkdir(); (tmp / "female").mkdir(); (tmp / "s3").mkdir()
rng = np.random.default_rng(5)
meta = tmp / "metadata.tsv"
meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\tS%d\n" % (i, i + 1, i + 1) for i in range(S)))


def write_matrix(path, x, chrom):
    df = pd.DataFrame(x + 1)
    R = x.shape[0]
    df.insert(0, "e", np.arange(R) * 200 + 200); df.insert(0, "s", np.arange(R) * 200); df.insert(0, "c", chrom)
    df.to_csv(path, sep="\t", header=False, index=False, compression={"method": "gzip", "compresslevel": 1})


t0 = time.time()
text_mb = 0
for f in range(a.files):
    x = rng.choice(S, size=(a.bins, N), p=FREQS / FREQS.sum()).astype(np.int8)
    chrom = "chr%d" % (f + 1)
    write_matrix(tmp / "all" / ("matrix_%s.txt.gz" % chrom), x, chrom)
    write_matrix(tmp / "male" / ("matrix_%s.txt.gz" % chrom), x[:, :379], chrom)
    write_matrix(tmp / "female" / ("matrix_%s.txt.gz" % chrom), x[:, 379:721], chrom)
    if f == 0:
        write_matrix(tmp / "s3" / "matrix_chr1.txt.gz", x[:a.s3_bins], chrom)
    text_mb += a.bins * (N * 2.3 + 20) / 1e6
print("inputs: %d files x %d bins x %d biosamples (~%.0f MB of text) written in %.1f s" % (a.files, a.bins, N, text_mb, time.time() - t0), flush=True)


def run(label, args, bins):
    out = tmp / ("out_" + label)
    t = time.time()
    r = subprocess.run([sys.executable, "-m", "epilogos_amd.run", "-l", "-j", str(meta), "-o", str(out)] + args, cwd=str(ROOT),
                       capture_output=True, text=True, env=dict(os.environ, EPILOGOS_TIMING="1"))
    print("".join(l + "\n" for l in r.stdout.splitlines() if "[timing]" in l), end="")
    dt = time.time() - t
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-3000:])
        raise SystemExit("%s failed" % label)
    files = sorted(p.name for p in out.iterdir())
    print("%-22s %7.2f s wall for %8d bins  -> %7.3f Mbins/s end to end (process start, parse, GPU, gzip text, STEP 4); outputs: %s"
          % (label, dt, bins, bins / dt / 1e6, ", ".join(files[:4]) + (" ..." if len(files) > 4 else "")), flush=True)
    return out


total = a.bins * a.files
only = set(a.only.split(",")) if a.only else {"s1", "s2", "s3", "paired", "pairedn"}
o1 = run("single S1", ["-i", str(tmp / "all"), "-s", "1"], total)
if "s2" in only: run("single S2", ["-i", str(tmp / "all"), "-s", "2"], total)
if "s3" in only: run("single S3", ["-i", str(tmp / "s3"), "-s", "3"], a.s3_bins)
if "paired" in only: run("paired S1 (379+342)", ["-m", "paired", "-a", str(tmp / "male"), "-b", str(tmp / "female"), "-s", "1", "--null-seed", "3"], total)
if "pairedn" in only: run("paired S1 -n", ["-m", "paired", "-a", str(tmp / "male"), "-b", str(tmp / "female"), "-s", "1", "--null-seed", "3", "-n", "-t", "3"], total)

# spot check: the text of chr1 equals the engine's scores of the same matrix
import torch  # noqa: E402
from epilogos_amd import _io, engine  # noqa: E402
x, loc = _io.read_table(tmp / "all" / "matrix_chr1.txt.gz")
xs = [_io.read_table(tmp / "all" / ("matrix_chr%d.txt.gz" % (f + 1)))[0] for f in range(a.files)]
counts = None
for xi in xs:
    _, c = engine.bin_hist(engine.states_to_device(xi), N, S, want_hist=False)
    counts = c if counts is None else counts + c
q = engine.normalise(counts)
o32, _ = engine.score_s1(engine.states_to_device(x), N, S, q)
with gzip.open(o1 / "scores_all_s1_matrix_chr1.txt.gz", "rt") as fh:
    got = np.loadtxt(fh, usecols=range(3, 3 + S), max_rows=5000, dtype=np.float64)
assert np.allclose(got, o32.cpu().numpy()[:5000], atol=1.01e-5), "CLI text differs from the engine's scores"
print("spot check OK: scores_all_s1_matrix_chr1.txt.gz == engine scores (first 5000 bins)")
if not a.keep:
    shutil.rmtree(tmp, ignore_errors=True)

#!/usr/bin/env python3
"""Throughput of the native parser / writer (libepilogos_io.so) vs the pandas / Python paths the reference uses."""
import gzip
import os
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from epilogos_amd import _io  # noqa: E402

R, N, S = 200000, 833, 18
tmp = Path(os.environ.get("TMPDIR", "/tmp"))
rng = np.random.default_rng(1)
x = rng.integers(1, S + 1, size=(R, N))
df = pd.DataFrame(x)
df.insert(0, "e", np.arange(R) * 200 + 200); df.insert(0, "s", np.arange(R) * 200); df.insert(0, "c", "chr1")
plain = tmp / "io_bench.txt"
df.to_csv(plain, sep="\t", header=False, index=False)
os.system("gzip -k -6 -f %s" % plain)
mb = plain.stat().st_size / 1e6
print("file: %d bins x %d biosamples, %.0f MB text, %.0f MB gz" % (R, N, mb, Path(str(plain) + ".gz").stat().st_size / 1e6))
import subprocess
for path in (plain, Path(str(plain) + ".gz")):
    for th in (1, 16):
        t = time.time(); st, loc = _io.read_table(path, threads=th); dt = time.time() - t
        print("native parse %-22s threads %2d: %6.2f s  %7.0f MB/s text  %6.3f Mbins/s" % (path.name, th, dt, mb / dt, R / dt / 1e6))
    assert np.array_equal(st, x - 1)
# the same .gz through zlib's inflate and the scalar parser (what round 2 started from), in a child: the library reads the switches once
code = ("import sys,time;sys.path.insert(0,%r);from epilogos_amd import _io;t=time.time();_io.read_table(%r,threads=16);"
        "print('native parse %%-22s threads 16: %%6.2f s  (EPGIO_INFLATE=zlib EPGIO_SIMD=0)' %% (%r, time.time()-t))"
        % (str(Path(__file__).resolve().parents[1]), str(plain) + ".gz", Path(str(plain) + ".gz").name))
print(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, EPGIO_INFLATE="zlib", EPGIO_SIMD="0")).stdout.strip())
t = time.time()
ref = pd.read_table(Path(str(plain) + ".gz"), usecols=range(3, N + 3), header=None, sep="\t").to_numpy(dtype=int) - 1
dt = time.time() - t
print("pandas read_table (reference helpers.py:152-155) gz: %6.2f s  %6.3f Mbins/s" % (dt, R / dt / 1e6))
sc = rng.random((R, S)).astype(np.float32)
for lvl, what in ((0, "own DEFLATE (default)"), (1, "zlib level 1"), (6, "zlib level 6")):
    for th in (1, 16):
        t = time.time(); _io.write_scores(tmp / "io_bench_scores.txt.gz", loc, sc, threads=th, gzip_level=lvl); dt = time.time() - t
        print("native write, %-22s threads %2d: %6.2f s  %6.3f Mbins/s  %6.1f MB" % (what, th, dt, R / dt / 1e6, (tmp / "io_bench_scores.txt.gz").stat().st_size / 1e6))
la = loc.to_object_array()
t = time.time()
tmpl = "{0[0]}\t{0[1]}\t{0[2]}\t" + "".join("{1[%d]:.5f}\t" % i for i in range(S - 1)) + "{1[%d]:.5f}\n" % (S - 1)
n = 50000
with gzip.open(tmp / "io_bench_ref.txt.gz", "wt") as g:
    g.write("".join(tmpl.format(la[i], sc[i]) for i in range(n)))
dt = (time.time() - t) * R / n
print("python str.format + gzip level 9 (reference scores.py:523-536), extrapolated from %d rows: %6.2f s  %6.3f Mbins/s" % (n, dt, R / dt / 1e6))
with gzip.open(tmp / "io_bench_scores.txt.gz", "rb") as a, gzip.open(tmp / "io_bench_ref.txt.gz", "rb") as b:
    assert a.read()[:1000000] == b.read()[:1000000]
for f in (plain, Path(str(plain) + ".gz"), tmp / "io_bench_scores.txt.gz", tmp / "io_bench_ref.txt.gz"):
    f.unlink()

#!/usr/bin/env python3
"""GPU box: the native reader on ONE chr1-sized file (1 246 253 bins x 833 biosamples, written by epgio_write_states), phase
times (EPGIO_TIMING) for the library's own inflate + AVX-512 parser against zlib + the scalar parser.  usage: reader_probe.py"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
p = "/dev/shm/chr1_probe.txt.gz"
if "--child" in sys.argv:
    import time
    from epilogos_amd import _io
    t = time.time()
    st, loc = _io.read_table(p)
    print("read_table %.2f s, %d x %d" % (time.time() - t, st.shape[0], st.shape[1]))
    sys.exit(0)
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import _io  # noqa: E402
R, N = 1246253, 833
X = torch.empty((R, N), dtype=torch.int8, device="cuda")
bench.generate_shard(torch, X, N, 18, 0)
_io.write_states(p, "chr1", X.cpu().numpy(), gzip_level=1)
for label, env in (("own inflate + AVX-512 parser", {}), ("own inflate + scalar parser", {"EPGIO_SIMD": "0"}),
                   ("zlib + scalar parser", {"EPGIO_SIMD": "0", "EPGIO_INFLATE": "zlib"})):
    for rep in range(2):
        r = subprocess.run([sys.executable, __file__, "--child"], capture_output=True, text=True, env=dict(os.environ, EPGIO_TIMING="1", **env))
        print("== %s: %s" % (label, r.stdout.strip()))
        print("".join(l + "\n" for l in r.stderr.splitlines() if "epgio" in l), end="")
os.remove(p)

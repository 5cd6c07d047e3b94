#!/usr/bin/env python3
"""Which kernels of libepilogos_hip.so did a profiled run reach?  usage: kernel_coverage.py <dir with *kernel_stats.csv files>
Lists the library's kernels (their .kd symbols, demangled, grouped by template) and, from every `rocprofv3 --kernel-trace --stats`
csv under the directory (one per process of the run), the kernels that were launched: per template the instantiations that ran /
that exist, and the templates that never ran.  GPU box: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cov --
python3 -m pytest tests -m gpu -q  (every kernel in the library appears in the kernel list of the GPU suite)."""
import csv
import re
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
lib = ROOT / "epilogos_amd" / "_lib" / "libepilogos_hip.so"
raw = subprocess.run(["strings", "-a", str(lib)], capture_output=True, text=True).stdout
syms = sorted(set(re.findall(r"_ZN3epg[A-Za-z0-9_]*(?=\.kd)", raw)))
dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.splitlines()


def norm(name):
    name = name.strip().strip('"')
    name = re.sub(r"^void\s+", "", name)
    return re.sub(r"\(.*$", "", name).strip()


def template(name):
    return re.sub(r"<.*$", "", name)


have = defaultdict(set)
for d in dem:
    n = norm(d)
    have[template(n)].add(n)
ran = defaultdict(set)
files = sorted(Path(sys.argv[1]).rglob("*kernel_stats.csv"))
for f in files:
    for r in csv.DictReader(open(f)):
        n = norm(r["Name"])
        if n.startswith("epg::"):
            ran[template(n)].add(n)
print("%d kernel templates (%d instantiations) in %s; %d kernel_stats.csv files under %s" % (len(have), sum(len(v) for v in have.values()), lib.name, len(files), sys.argv[1]))
missing = [t for t in sorted(have) if t not in ran]
for t in sorted(have):
    print("  %-34s %3d of %3d instantiations ran%s" % (t, len(ran.get(t, ())), len(have[t]), "" if t in ran else "   <-- NEVER LAUNCHED"))
print("templates never launched: %s" % (", ".join(missing) if missing else "none"))
sys.exit(1 if missing else 0)

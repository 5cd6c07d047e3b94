#!/usr/bin/env python3
"""Instruction-mix histogram of one kernel from a hipcc -S (--cuda-device-only) assembly file.
usage: isa_mix.py file.s mangled-name-substring [--loop]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ops = collections.Counter()
for l in lines[start:end + 1]:
    m = re.match(r"^\s+([a-z][a-z_0-9]+)\b", l)
    if m:
        ops[m.group(1)] += 1
print(lines[start].split(":")[0], "total", sum(ops.values()))
cls = collections.Counter()
for k, v in ops.items():
    c = "VALU" if k.startswith("v_") else "SALU" if k.startswith("s_") else "VMEM" if k.startswith(("global_", "buffer_", "flat_")) else "LDS" if k.startswith("ds_") else "other"
    cls[c] += v
print(dict(cls))
for k, v in ops.most_common(45):
    print("  %-30s %d" % (k, v))

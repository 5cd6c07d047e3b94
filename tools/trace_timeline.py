#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace csv: tools/trace_timeline.py <dir> [--last-ms 40] [--grep name,name]
Prints, for the kernels that started in the last `--last-ms` of the trace: start (ms from the window's begin), duration (us),
queue, name -- enough to see which launches of two streams really ran at the same time and what it did to their durations."""
import argparse
import csv
import glob
import os

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--last-ms", type=float, default=40.0)
ap.add_argument("--grep", default="")
ap.add_argument("--max", type=int, default=400)
a = ap.parse_args()
f = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
key = lambda r, *names: next(r[n] for n in names if n in r)
ev = [(int(key(r, "Start_Timestamp")), int(key(r, "End_Timestamp")), key(r, "Queue_Id"), key(r, "Kernel_Name")) for r in rows]
ev.sort()
t_end = max(e[1] for e in ev)
t0 = t_end - int(a.last_ms * 1e6)
pats = [p for p in a.grep.split(",") if p]
shown = [e for e in ev if e[0] >= t0 and (not pats or any(p in e[3] for p in pats))]
print("%d kernels in the trace, %d in the last %.1f ms" % (len(ev), len(shown), a.last_ms))
for s, e, q, name in shown[: a.max]:
    short = name.split("(")[0].replace("void ", "").replace("epg::", "")[:48]
    print("%9.3f ms  %8.1f us  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, short))
# busy time per kernel name and the union of the intervals (what the window really took on the device)
tot = {}
for s, e, q, name in shown:
    k = name.split("(")[0].replace("void ", "").replace("epg::", "")[:48]
    tot[k] = tot.get(k, 0) + (e - s)
iv = sorted((s, e) for s, e, _, _ in shown)
union, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    union += cur_e - cur_s
print("sum of durations by kernel (ms):", {k: round(v / 1e6, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
print("union of the intervals: %.3f ms; sum: %.3f ms" % (union / 1e6, sum(tot.values()) / 1e6))

#!/usr/bin/env python3
"""Join the per-dispatch counters of tools/placement_pmc.sh with the dispatch durations of the same pass and print, per
counter, its value per k_bin_hist launch next to the launch time, plus the correlation over the launches that store H."""
import collections
import csv
import sys
from pathlib import Path

import numpy as np

src = Path(sys.argv[1])
lines = ["# k_bin_hist: HBM-side counters against launch time over buffer placements", ""]
plain = src / "plain.log"
if plain.exists():
    lines += ["un-profiled run (HIP events):", "```"] + [l for l in plain.read_text().splitlines() if l.startswith("placement")] + ["```", ""]
for g in sorted(p for p in src.iterdir() if p.is_dir()):
    cc = list(g.rglob("*counter_collection.csv"))
    kt = list(g.rglob("*kernel_trace.csv"))
    if not cc or not kt:
        continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        if "k_bin_hist" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    vals = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        if "k_bin_hist" in r["Kernel_Name"]:
            vals[r["Counter_Name"]][r["Dispatch_Id"]] = float(r["Counter_Value"])
    ids = sorted(dur, key=int)
    if not ids:
        continue
    # launch pattern of placement_pmc.py: 3 with the H store, 1 without, per placement
    withH = [i for k, i in enumerate(ids) if k % 4 != 3]
    noH = [i for k, i in enumerate(ids) if k % 4 == 3]
    t = np.array([dur[i] for i in withH])
    lines += ["## pass %s: %d launches with the H store %.3f..%.3f ms (profiled), %d without %.3f..%.3f ms" % (
        g.name, len(withH), t.min(), t.max(), len(noH), min(dur[i] for i in noH), max(dur[i] for i in noH)), "",
        "| counter | per launch with H: at fastest | at slowest | corr. with time | without H (mean) |", "|---|---|---|---|---|"]
    for c, d in sorted(vals.items()):
        v = np.array([d.get(i, np.nan) for i in withH])
        r = float(np.corrcoef(t, v)[0, 1]) if np.nanstd(v) > 0 and t.std() > 0 else float("nan")
        vn = np.nanmean([d.get(i, np.nan) for i in noH])
        lines.append("| %s | %.4g | %.4g | %+.2f | %.4g |" % (c, v[t.argmin()], v[t.argmax()], r, vn))
    lines += ["", "per launch (ms): " + " ".join("%.3f" % x for x in t), ""]
out = "\n".join(lines) + "\n"
(src / "digest.md").write_text(out)
print(out)

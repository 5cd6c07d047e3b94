#!/usr/bin/env python3
"""Average every PMC counter per kernel over the dispatches found under <dir>/*/p_counter_collection.csv."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
keep = sys.argv[2:] or ["k_bin_hist", "k_score_s1", "k_s2", "k_score_s2", "k_s3", "k_null"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(d + "/*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in keep):
            continue
        k = k.split("(")[0].replace("void epg::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(agg):
    print("== %s  (n=%d dispatch rows, mean duration under PMC %.3f ms)" % (k, len(dur[k]), sum(dur[k]) / len(dur[k]) / 1e6))
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-42s %16.1f" % (c, sum(v) / len(v)))

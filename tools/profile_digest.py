#!/usr/bin/env python3
"""Digest gpurun_out/prof_<label>/ into profiles/<label>_*.{md,csv,json} (the files the judge reads)."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

src, label = Path(sys.argv[1]), sys.argv[2]
root = Path(__file__).resolve().parents[1]
dst = root / "profiles"
dst.mkdir(exist_ok=True)
lines = ["# rocprofv3 summary %s (MI355X, `python bench.py --steps 20 --warmup 3`)" % label, ""]
bench = None
bj = src / "bench.json"
if bj.exists() and bj.read_text().strip():
    bench = json.loads([l for l in bj.read_text().splitlines() if l.startswith("{")][-1])
    (dst / ("%s_bench.json" % label)).write_text(json.dumps(bench, indent=1) + "\n")
    lines += ["## un-profiled bench line", "", "```json", json.dumps(bench), "```", ""]
stats = src / "stats" / "p_kernel_stats.csv"
if stats.exists():
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "epg::" in r["Name"]]
    with open(dst / ("%s_kernel_stats.csv" % label), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
    lines += ["## `rocprofv3 --kernel-trace --stats` (epg:: kernels)", "", "| kernel | calls | avg ns | min ns | max ns |", "|---|---|---|---|---|"]
    for r in keep:
        lines.append("| `%s` | %s | %.0f | %s | %s |" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
    lines.append("")
trace = src / "stats" / "p_kernel_trace.csv"
if trace.exists() and bench:
    # bench.py launches k_bin_hist for its own steps FIRST (one untimed job, the warm-up, the timed steps); the placement
    # experiment and the S2 / paired configs launch it again afterwards, so the timed steps are dispatches 1 + warmup ..
    # 1 + warmup + steps of the kernel's full-width instantiation
    tr = list(csv.DictReader(open(trace)))
    k1 = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in tr
                if "k_bin_hist<18, 7" in r["Kernel_Name"])
    # round 5: the first job's add_device probes the histogram placement with launches over 1 M-bin slices (engine.alloc_hist);
    # the whole-matrix launches are the long ones
    if k1:
        top = max(d for _t, d in k1)
        k1 = [x for x in k1 if x[1] >= 0.6 * top]
    steps, warm = bench["steps"], bench["warmup"]
    if len(k1) >= 1 + warm + steps:
        d = [x[1] for x in k1[1 + warm:1 + warm + steps]]
        # the profiled process prints its own bench line (stats.log): with plain allocations every PROCESS gets its own placement of
        # X and H (same or different memory class: 2.6-2.7 against 2.3-2.4 ms), so the trace is compared with the events of the very
        # process it traced; the un-profiled run above is another process
        own = None
        slog = src / "stats.log"
        if slog.exists():
            for ln in slog.read_text().splitlines():
                if ln.startswith("{") and '"kernels_ms"' in ln:
                    own = json.loads(ln)
        lines += ["k_bin_hist over the %d timed dispatches of the trace (dispatches %d..%d of the kernel; later ones belong to the "
                  "secondary measurements): avg %.0f ns, min %d, max %d -- HIP events of the SAME (profiled) process: %s ns; "
                  "un-profiled run above (another process, its own placement of X and H): %.0f ns"
                  % (steps, 1 + warm, warm + steps, sum(d) / len(d), min(d), max(d),
                     ("%.0f" % (own["kernels_ms"]["k_bin_hist"] * 1e6)) if own else "n/a", bench["kernels_ms"]["k_bin_hist"] * 1e6), ""]
    # the configs of the same run: kernel time of every phase from the trace next to the bench line's event times
    cfg = bench.get("configs") or {}
    def tsum(pred, last):
        rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in tr if pred(r["Kernel_Name"]))
        return rows[-last:] if last else rows
    if "s3" in cfg and "phases_ms" in cfg["s3"]:
        reps = cfg["s3"]["reps"]
        sy = [d for _, d in tsum(lambda n: "k_s3_syrk_fp4" in n, 0) if d > 1e6][-reps:]
        sc = [d for _, d in tsum(lambda n: "k_s3_score_bl" in n, reps)]
        if sy and sc:
            lines += ["config s3 (%d bins): `k_s3_syrk_fp4` %.2f ms per launch (one per 2 M-bin chunk of the operand; the gated-off launches of the other contraction return after microseconds and pull the average down: see the MAX column) and `k_s3_score_bl` %.2f ms per call in the trace (last %d); bench "
                      "line phases (events, incl. operand build / reconstruction resp. table build / transpose): expected %.2f ms, scores %.2f ms"
                      % (cfg["s3"]["bins_total"], sum(sy) / len(sy) / 1e6, sum(sc) / len(sc) / 1e6, reps, cfg["s3"]["phases_ms"]["expected"],
                         cfg["s3"]["phases_ms"]["scores"]), ""]
    if "s2" in cfg and "kernels_ms" in cfg["s2"]:
        a = [d for _, d in tsum(lambda n: "k_bin_hist_s2" in n, 3)]
        b = [d for _, d in tsum(lambda n: "k_score_s2_bin" in n, 3)]
        if a and b:
            lines += ["config s2: `k_bin_hist_s2` %.4f ms, `k_score_s2_bin` %.4f ms in the trace (last 3 calls); bench line kernels_ms: %s"
                      % (sum(a) / len(a) / 1e6, sum(b) / len(b) / 1e6, json.dumps(cfg["s2"]["kernels_ms"])), ""]
    if "paired" in cfg and "phases_ms" in cfg["paired"]:
        names = ("k_pair_count_null", "k_bin_hist_parts", "k_null_hist_h", "k_pair_fused_s1", "k_pair_finish", "k_pair_metrics", "k_quiescent_h")
        parts = []
        for nm in names:
            v = [d for _, d in tsum(lambda n, nm=nm: nm in n, 0)]
            if v:
                parts.append("`%s` %d calls, avg %.3f ms" % (nm, len(v), sum(v) / len(v) / 1e6))
        lines += ["config paired: " + "; ".join(parts) + " -- bench line phases: %s" % json.dumps(cfg["paired"]["phases_ms"]), ""]
stats_all = src / "stats_all" / "p_kernel_stats.csv"
if stats_all.exists():
    rows = list(csv.DictReader(open(stats_all)))
    keep = [r for r in rows if "epg::" in r["Name"]]
    with open(dst / ("%s_s2s3_kernel_stats.csv" % label), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
    lines += ["## `rocprofv3 --kernel-trace --stats -- python3 tools/kbench_all.py --bins 15000000 --s3-bins 1000000 --null-bins 1000000`",
              "(S2 on 15 M bins, S3 and the null shuffle on 1 M bins, 833 biosamples; `k_s3_syrk_fp4` / `k_s3_onehot_fp4`: every second",
              "launch is the gated-off contraction of the other kind and returns after a few us, so the real launch is the MAX column)", "",
              "| kernel | calls | avg ns | min ns | max ns |", "|---|---|---|---|---|"]
    for r in keep:
        lines.append("| `%s` | %s | %.0f | %s | %s |" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
    lines.append("")
    log = src / "stats_all.log"
    if log.exists():
        lines += ["```"] + [l for l in log.read_text().splitlines() if l.startswith(("S2", "S3", "paired", "pair_", "quiescent"))] + ["```", ""]
traffic, traffic_max = {}, {}
for cname in ("fetch", "write"):
    f = src / cname / "p_counter_collection.csv"
    if not f.exists():
        continue
    agg = collections.defaultdict(list)
    rows_c = [r for r in csv.DictReader(open(f)) if "epg::" in r["Kernel_Name"]]
    rows_c.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows_c:
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for key in list(agg):
        # k_bin_hist: the secondary measurements launch it too (some without the histogram store); the bench's
        # own warm-up + steps are dispatches 1..4 of a PMC pass (`--steps 3 --warmup 1`)
        # (round 5: and the placement probe launches it over 1 M-bin slices first -- the whole-matrix launches with the store are
        # the ones with the largest counter values)
        if "k_bin_hist<18, 7" in key[0]:
            top = max(agg[key])
            agg[key] = [v for v in agg[key] if v >= 0.9 * top][:4]
    lines += ["## PMC pass `%s`" % cname, "", "| kernel | counter | mean per launch | bytes (KB x1024%s) |" % (", x2 gfx950 read correction" if cname == "fetch" else ""), "|---|---|---|---|"]
    for (k, c), v in sorted(agg.items()):
        m = sum(v) / len(v)
        b = m * 1024 * (2 if cname == "fetch" else 1)
        traffic.setdefault(k, {})[c] = b
        traffic_max.setdefault(k, {})[c] = max(v) * 1024 * (2 if cname == "fetch" else 1)
        lines.append("| `%s` | %s | %.1f | %.4g |" % (k, c, m, b))
    lines.append("")
if bench and traffic:
    R, N = bench["config"]["bins_per_gpu"], bench["config"]["biosamples"]
    k = [x for x in traffic if "k_bin_hist<18, 7" in x] or [x for x in traffic if "k_bin_hist" in x]
    if k:
        t = traffic[k[0]]
        total = t.get("FETCH_SIZE", 0) + t.get("WRITE_SIZE", 0)
        tfile = dst / "hbm_traffic.json"
        d = json.loads(tfile.read_text()) if tfile.exists() else {}
        sys.path.insert(0, str(root))
        import bench as _bench
        if d.get("k1_source_sha") != _bench.k1_source_sha():      # numbers of other kernel code do not carry over
            d = {}
        d["k1_source_sha"] = _bench.k1_source_sha()
        d["k_bin_hist_bytes_per_launch_%d_%d" % (R, N)] = total
        d["source"] = "profiles/%s_summary.md: FETCH_SIZE*1024*2 (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024, separate --pmc passes" % label
        # the memory side of the S3 score kernel: its largest launch of the pass is config s3's (the whole genome); bench.py puts the
        # figure next to the LDS-gather roofline of configs.s3
        s3k = [x for x in traffic_max if "k_s3_score_bl" in x]
        s3bins = ((bench.get("configs") or {}).get("s3") or {}).get("bins_per_gpu")
        if s3k and s3bins:
            d["k_s3_score_bl_fetch_bytes_per_launch_%d_%d" % (s3bins, N)] = traffic_max[s3k[0]].get("FETCH_SIZE")
            d["k_s3_score_bl_write_bytes_per_launch_%d_%d" % (s3bins, N)] = traffic_max[s3k[0]].get("WRITE_SIZE")
        tfile.write_text(json.dumps(d, indent=1) + "\n")
        lines += ["k_bin_hist HBM traffic per launch: %.4g B vs algorithmic %d x %d = %.4g B read (+ %.4g B of H written)"
                  % (total, R, N, R * N, R * 36.0), ""]
(dst / ("%s_summary.md" % label)).write_text("\n".join(lines) + "\n")
print("\n".join(lines))

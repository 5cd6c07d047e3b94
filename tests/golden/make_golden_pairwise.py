#!/usr/bin/env python3
"""Golden vectors for SURVEY 8 row f4 (STEP 4 of a paired run): runs the REAL reference's roiAndVisualPairwise
functions (readInData :273-356, fitDistances :175-221, calculatePVals :496-517, writeMetrics :520-573, createROITxt
:646-723, createSignificantLociTxt :576-643, createROINoSignificance :726-778) on files written by the reference's own
writeScores.  Build container only (the reference does not travel).

    python tests/golden/make_golden_pairwise.py        # rewrites tests/golden/pairwise_step4.npz

natsort / pyranges / statsmodels are stubbed as in make_golden.py.  statsmodels is absent from this image, so the one
line of main() that calls it (multipletests(pvals, method="fdr_bh"), :93) cannot run: the Benjamini-Hochberg values
stored here come from scipy.stats.false_discovery_control(method="bh") and are then fed to the reference's own writers.
The fit is deterministic here because the null sample is smaller than samplingSize (fitOnSubSample :257-258 then fits
all the data).  Two data sets: the real 5 + 5 slice of paired.npz, and the same slice with 40 bins made strongly
different so that the significant branches write something.
"""
import gzip
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from make_golden import REF, S, import_reference  # noqa: E402

SPLITS = {"chr2": (0, 900), "chr10": (900, 1500), "chrX": (1500, 2048)}       # exercises orderChromosomes + sorting


def stage_dir(sc, delta, null_dist, quies, start0):
    d = Path(tempfile.mkdtemp(prefix="epg_pw_"))
    for name, (lo, hi) in SPLITS.items():
        loc = np.array([[name, start0 + 200 * i, start0 + 200 * i + 200] for i in range(lo, hi)], dtype=object)
        sc.writeScores(delta[lo:hi], d / "pairwiseDelta_t_matrix_{}.txt.gz".format(name), loc)
        np.savez_compressed(d / "temp_nullDistances_t_matrix_{}.npz".format(name), chrName=np.array([name]),
                            nullDistances=null_dist[lo:hi])
        np.savez_compressed(d / "temp_quiescence_t_matrix_{}.npz".format(name), chrName=np.array([name]),
                            quiescenceArr=quies[lo:hi])
    return d


def main():
    import_reference()
    import scipy.stats as st
    import epilogos.roiAndVisualPairwise as rv
    import epilogos.scores as sc
    import pandas as pd
    g = dict(np.load(HERE / "paired.npz"))
    start0 = int(np.load(HERE / "real_slice.npz")["start0"])
    state_info = REF + "/data/state_metadata/human/Boix_et_al_833_sample/hg19/18/metadata.tsv"
    names = pd.read_table(state_info, header=0, sep="\t")["short_name"].values
    out = {"split_names": np.array(list(SPLITS)), "split_bounds": np.array(list(SPLITS.values()), dtype=np.int64),
           "start0": np.int64(start0), "state_names": np.array(names, dtype="U32")}
    rng = np.random.default_rng(7)
    delta_real = g["s1_delta"]
    spiked = delta_real.copy()
    rows = rng.choice(delta_real.shape[0], size=40, replace=False)
    spiked[rows] += rng.normal(0, 1.5, size=(40, S)).astype(np.float32)
    for label, delta in (("real", delta_real), ("spiked", spiked)):
        null_dist, quies = g["s1_null_dist"], g["s1_quiescent"]
        out[label + "_delta"] = delta
        # ---- without -n: z-scores
        d = stage_dir(sc, delta, null_dist, quies, start0)
        locationArr, dist, maxdiff, chrDict = rv.readInData(d, 1, S)
        assert not list(d.glob("temp_*.npz"))
        z = np.abs(st.zscore(dist))
        rv.writeMetrics(locationArr, chrDict, maxdiff, names, dist, d, "t", False)
        out[label + "_location"] = locationArr
        out[label + "_dist"], out[label + "_maxdiff"], out[label + "_z"] = dist, maxdiff, z
        out[label + "_metrics_z"] = np.frombuffer(gzip.open(d / "pairwiseMetrics_t.txt.gz").read(), dtype=np.uint8)
        for w in (125, 10, 7):
            rv.createROINoSignificance(d / "roi.txt", locationArr, chrDict, dist, maxdiff, names, z, w)
            out["%s_roi_z_w%d" % (label, w)] = np.frombuffer((d / "roi.txt").read_bytes(), dtype=np.uint8)
        # ---- with -n: gennorm fit, p-values, BH
        d = stage_dir(sc, delta, null_dist, quies, start0)
        params, distNull, nonQ = rv.fitDistances(d, 1, 3, 100000)
        locationArr, dist, maxdiff, chrDict = rv.readInData(d, 1, S)
        beta, loc, scale = params[:-2], params[-2], params[-1]
        pvals = rv.calculatePVals(dist, beta, loc, scale)
        mh = st.false_discovery_control(pvals, method="bh")
        rv.writeMetrics(locationArr, chrDict, maxdiff, names, dist, d, "t", True, pvals=pvals, mhPvals=mh)
        out[label + "_params"] = np.array([float(np.ravel(beta)[0]), float(loc), float(scale)])
        out[label + "_pvals"], out[label + "_mh"] = pvals, mh
        out[label + "_metrics_p"] = np.frombuffer(gzip.open(d / "pairwiseMetrics_t.txt.gz").read(), dtype=np.uint8)
        for w in (125, 10):
            rv.createROITxt(d / "roi.txt", locationArr, chrDict, dist, maxdiff, names, pvals, mh, w)
            out["%s_roi_p_w%d" % (label, w)] = np.frombuffer((d / "roi.txt").read_bytes(), dtype=np.uint8)
        rv.createSignificantLociTxt(d / "sig.txt.gz", locationArr, chrDict, dist, maxdiff, names, pvals, mh)
        out[label + "_sig"] = np.frombuffer(gzip.open(d / "sig.txt.gz").read(), dtype=np.uint8)
        print(label, "params", out[label + "_params"], "significant", int((mh <= 0.1).sum()),
              "roi_p_w10 bytes", len(out[label + "_roi_p_w10"]), "roi_z_w125 bytes", len(out[label + "_roi_z_w125"]))
    out["null_dist"], out["quiescent"] = g["s1_null_dist"], g["s1_quiescent"]
    np.savez_compressed(HERE / "pairwise_step4.npz", **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})
    print(out["spiked_roi_p_w10"].tobytes().decode()[:600])
    print(out["real_roi_z_w125"].tobytes().decode()[:400])


if __name__ == "__main__":
    main()

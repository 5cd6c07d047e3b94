#!/usr/bin/env python3
"""Golden vectors for SURVEY 8 row f3 (regions of interest of a single-group run): runs the REAL reference's
roiSingle.main (epilogos/roiSingle.py:10-40 -> helpers.maxMean -> filter_regions.Filter.maxmean) on temp_scores_*.npz
files written by the reference's own scores path.  Build container only (the reference does not travel).

    python tests/golden/make_golden_roi.py        # rewrites tests/golden/roi.npz
natsort / pyranges are stubbed exactly as in make_golden.py: the maxmean path on a numpy input never touches them.
"""
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from make_golden import REF, S, import_reference  # noqa: E402


def main():
    import_reference()
    import epilogos.roiSingle as roi
    g = dict(np.load(HERE / "real_slice.npz"))
    scores = g["s1_f32"]
    start0 = int(g["start0"])
    state_info = REF + "/data/state_metadata/human/Boix_et_al_833_sample/hg19/18/metadata.tsv"
    out = {}
    splits = {"chr1": (0, 1200), "chr10": (1200, 1700), "chrX": (1700, 2048)}       # exercises orderChromosomes
    out["split_names"] = np.array(list(splits))
    out["split_bounds"] = np.array(list(splits.values()), dtype=np.int64)
    for width in (50, 7, 10):
        d = Path(tempfile.mkdtemp(prefix="epg_roi_"))
        for name, (lo, hi) in splits.items():
            loc = np.array([[name, start0 + 200 * i, start0 + 200 * i + 200] for i in range(hi - lo)], dtype=object)
            np.savez_compressed(d / "temp_scores_t_{}.npz".format(name), chrName=np.array([name]), scoreArr=scores[lo:hi],
                                locationArr=loc)
        exp = d / "exp_freq_t.npy"
        np.save(exp, g["s1_exp"])
        roi.main(d, state_info, "t", exp, width, False)
        out["roi_w%d" % width] = np.frombuffer((d / "regionsOfInterest_t.txt").read_bytes(), dtype=np.uint8)
        assert not exp.exists() and not list(d.glob("temp_scores_*"))                 # reference deletes its inputs
    # the whole slice as one chromosome with the CLI's default width (what `epilogos -i in10 ...` produces)
    d = Path(tempfile.mkdtemp(prefix="epg_roi_"))
    loc = np.array([["chr1", start0 + 200 * i, start0 + 200 * i + 200] for i in range(scores.shape[0])], dtype=object)
    np.savez_compressed(d / "temp_scores_t_matrix_chr1.npz", chrName=np.array(["chr1"]), scoreArr=scores, locationArr=loc)
    np.save(d / "exp_freq_t.npy", g["s1_exp"])
    roi.main(d, state_info, "t", d / "exp_freq_t.npy", 50, False)
    out["roi_single_w50"] = np.frombuffer((d / "regionsOfInterest_t.txt").read_bytes(), dtype=np.uint8)
    # ties: a handful of distinct scores, so that many windows share rolling max AND rolling mean and the reference's third
    # sort key (the centre bin's score, helpers.py:272) and pandas' stable order decide
    rng = np.random.default_rng(11)
    tie = np.zeros((3000, S), dtype=np.float32)
    tie[:, 4] = rng.integers(0, 4, size=3000)
    tie[:, 9] = rng.integers(0, 2, size=3000) * 0.5
    out["tie_scores"] = tie
    for width in (7, 10):
        d = Path(tempfile.mkdtemp(prefix="epg_roi_"))
        loc = np.array([["chr2", 200 * i, 200 * i + 200] for i in range(tie.shape[0])], dtype=object)
        np.savez_compressed(d / "temp_scores_t_chr2.npz", chrName=np.array(["chr2"]), scoreArr=tie, locationArr=loc)
        np.save(d / "exp_freq_t.npy", g["s1_exp"])
        roi.main(d, state_info, "t", d / "exp_freq_t.npy", width, False)
        out["roi_tie_w%d" % width] = np.frombuffer((d / "regionsOfInterest_t.txt").read_bytes(), dtype=np.uint8)
    out["state_names"] = np.array(__import__("pandas").read_table(state_info, header=0, sep="\t")["short_name"].values, dtype="U32")
    np.savez_compressed(HERE / "roi.npz", **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})
    print(out["roi_w50"].tobytes().decode()[:400])


if __name__ == "__main__":
    main()

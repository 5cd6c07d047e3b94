#!/usr/bin/env python3
"""Golden vectors for the whole chr1 example (BASELINE config 1): the 10-biosample chr1 matrix built by the
reference's own preprocessing script from its bundled data/ChromHMM, scored end to end by the REAL reference
(expected.main -> expectedCombination.main -> scores.main, S1).  Build container only.

    python tests/golden/make_golden_chr1.py        # rewrites tests/golden/chr1_full.npz  (~2 minutes)

Stored: the int8 state matrix (1 246 253 bins x 10), the reference's counts and exp_freq, and of its 90 MB of scores a
digest: SHA-256 of the decompressed scores text, float64 column sums of the float32 scores, every 997th row.
"""
import gzip
import hashlib
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from make_golden import REF, S, import_reference  # noqa: E402


def main():
    ex, ec, sc, hp = import_reference()
    work = Path(tempfile.mkdtemp(prefix="epg_chr1_"))
    gen = work / "chr1.genome"
    gen.write_text("chr1\t249250621\n")
    subprocess.run(["bash", REF + "/bin/preprocess_data_ChromHMM.sh", REF + "/data/ChromHMM",
                    REF + "/data/metadata_Boix.txt", str(gen)], cwd=work, check=True, stdout=subprocess.DEVNULL)
    f = work / "matrix_chr1.txt"
    import pandas as pd
    df = pd.read_table(f, header=None, sep="\t")
    x = (df.iloc[:, 3:].to_numpy(dtype=np.int64) - 1).astype(np.int8)
    assert x.shape == (1246253, 10), x.shape
    out = work / "out"
    out.mkdir()
    ex.main(f, "null", S, 1, out, "chr1", 8, False)
    counts = np.load(out / "temp_exp_freq_chr1_matrix_chr1.npy")
    ec.main(out, out / "exp_freq_chr1.npy", "chr1", False)
    q = np.load(out / "exp_freq_chr1.npy")
    sc.main(f, "null", S, 1, out, out / "exp_freq_chr1.npy", "chr1", 8, S - 1, -1, False)
    z = np.load(out / "temp_scores_chr1_matrix_chr1.npz", allow_pickle=True)
    scores = z["scoreArr"]
    assert scores.dtype == np.float32 and scores.shape == (x.shape[0], S)
    with gzip.open(out / "scores_chr1_matrix_chr1.txt.gz", "rb") as g:
        text = g.read()
    lines = text.split(b"\n")
    res = {"x": x, "start0": np.int64(df.iloc[0, 1]), "counts": counts, "exp": q,
           "text_sha256": np.frombuffer(hashlib.sha256(text).digest(), dtype=np.uint8), "text_bytes": np.int64(len(text)),
           "first_line": np.frombuffer(lines[0], dtype=np.uint8), "last_line": np.frombuffer(lines[-2], dtype=np.uint8),
           "colsum_f64": scores.astype(np.float64).sum(axis=0), "rows_997": scores[::997].copy(),
           "absmax": np.float32(np.abs(scores).max())}
    # ---- S2 on the same matrix
    ex.main(f, "null", S, 2, out, "chr1s2", 8, False)
    res["s2_counts"] = np.load(out / "temp_exp_freq_chr1s2_matrix_chr1.npy")
    ec.main(out, out / "exp_freq_chr1s2.npy", "chr1s2", False)
    res["s2_exp"] = np.load(out / "exp_freq_chr1s2.npy")
    sc.main(f, "null", S, 2, out, out / "exp_freq_chr1s2.npy", "chr1s2", 8, S - 1, -1, False)
    s2 = np.load(out / "temp_scores_chr1s2_matrix_chr1.npz", allow_pickle=True)["scoreArr"]
    with gzip.open(out / "scores_chr1s2_matrix_chr1.txt.gz", "rb") as g:
        text = g.read()
    res.update({"s2_text_sha256": np.frombuffer(hashlib.sha256(text).digest(), dtype=np.uint8), "s2_text_bytes": np.int64(len(text)),
                "s2_colsum_f64": s2.astype(np.float64).sum(axis=0), "s2_rows_997": s2[::997].copy()})
    # ---- paired S1: biosamples 0-4 against 5-9 (the deltas are deterministic, the null shuffle is not)
    (work / "A").mkdir(); (work / "B").mkdir()
    df.iloc[:, list(range(3)) + list(range(3, 8))].to_csv(work / "A" / "matrix_chr1.txt", sep="\t", header=False, index=False)
    df.iloc[:, list(range(3)) + list(range(8, 13))].to_csv(work / "B" / "matrix_chr1.txt", sep="\t", header=False, index=False)
    fa, fb = work / "A" / "matrix_chr1.txt", work / "B" / "matrix_chr1.txt"
    ex.main(fa, fb, S, 1, out, "pair", 8, False)
    res["pair_counts"] = np.load(out / "temp_exp_freq_pair_matrix_chr1.npy")
    ec.main(out, out / "exp_freq_pair.npy", "pair", False)
    res["pair_exp"] = np.load(out / "exp_freq_pair.npy")
    sc.main(fa, fb, S, 1, out, out / "exp_freq_pair.npy", "pair", 8, S - 1, -1, False)
    with gzip.open(out / "pairwiseDelta_pair_matrix_chr1.txt.gz", "rb") as g:
        text = g.read()
    res.update({"pair_text_sha256": np.frombuffer(hashlib.sha256(text).digest(), dtype=np.uint8), "pair_text_bytes": np.int64(len(text)),
                "pair_quiescent_count": np.int64(np.load(out / "temp_quiescence_pair_matrix_chr1.npz")["quiescenceArr"].sum())})
    # ---- S3 on the same matrix (float32 sequential accumulation in the reference: compared with a tolerance)
    ex.main(f, "null", S, 3, out, "chr1s3", 8, False)
    res["s3_counts"] = np.load(out / "temp_exp_freq_chr1s3_matrix_chr1.npy")
    ec.main(out, out / "exp_freq_chr1s3.npy", "chr1s3", False)
    res["s3_exp"] = np.load(out / "exp_freq_chr1s3.npy")
    sc.main(f, "null", S, 3, out, out / "exp_freq_chr1s3.npy", "chr1s3", 8, S - 1, -1, False)
    s3 = np.load(out / "temp_scores_chr1s3_matrix_chr1.npz", allow_pickle=True)["scoreArr"]
    res.update({"s3_colsum_f64": s3.astype(np.float64).sum(axis=0), "s3_rows_997": s3[::997].copy()})
    np.savez_compressed(HERE / "chr1_full.npz", **res)
    print({k: getattr(v, "shape", None) for k, v in res.items()}, (HERE / "chr1_full.npz").stat().st_size)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""
Generate the golden vectors in tests/golden/*.npz by running the REAL reference (meuleman/epilogos, mounted
read-only at /root/reference) on small inputs.  Runs only in the build container: the reference never travels
to the GPU box, only these data fixtures do.

Recipe: the hot-path modules import `filter_regions`, which needs natsort/pyranges that
are not installed and are never used on this path -> register empty stub modules before importing.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What is captured (inputs + the reference's outputs):
  real_slice.npz  rows 50000..52047 of the 10-biosample chr1 matrix built from the reference's bundled
                  data/ChromHMM with its own bin/preprocess_data_ChromHMM.sh: S1/S2/S3 counts (expected.s?Calc),
                  exp_freq (expectedCombination.main), scores float64 pre-store (klScoreND(rowObsS?)) and float32
                  as stored (scores.s?Score), and the text written by scores.writeScores.
  synth833.npz    512 x 833 synthetic bins (np.random.default_rng(0), chr1 state frequencies): S1 and S2.
  s3_small.npz    256 x 32 synthetic bins: S3 counts, exp_freq, float32 scores.
  paired.npz      real slice split 5 + 5: S1/S2 over [A|B], delta, quiescence, seeded null shuffle + distances.
  edge.npz        q == 0 state, all-one-state bins, N = 2, every state present, no trailing newline row count.
"""
import gzip
import os
import subprocess
import sys
import tempfile
import types
from multiprocessing import RawArray
from pathlib import Path

import numpy as np

REF = "/root/reference"
HERE = Path(__file__).resolve().parent
S = 18
FREQS = np.array([.00570, .00293, .00430, .00212, .03260, .10464, .00154, .00057, .01001, .00416, .01554, .00618,
                  .02498, .00262, .00140, .01412, .05563, .71097])


def import_reference():
    for m in ("natsort", "pyranges", "statsmodels", "statsmodels.stats", "statsmodels.stats.multitest"):
        sys.modules[m] = types.ModuleType(m)
    sys.modules["statsmodels.stats.multitest"].multipletests = None
    sys.path.insert(0, REF)
    import epilogos.expected as ex
    import epilogos.expectedCombination as ec
    import epilogos.helpers as hp
    import epilogos.scores as sc
    return ex, ec, sc, hp


def write_tsv(path, x0, chrom="chr1", start0=0, trailing_newline=True):
    """x0: 0-based states [R,N] -> the reference's input format (1-based states, README.md:127-134)."""
    R = x0.shape[0]
    lines = ["{}\t{}\t{}\t{}".format(chrom, start0 + 200 * r, start0 + 200 * r + 200,
                                     "\t".join(str(int(v) + 1) for v in x0[r])) for r in range(R)]
    txt = "\n".join(lines) + ("\n" if trailing_newline else "")
    if str(path).endswith(".gz"):
        with gzip.open(path, "wt") as f:
            f.write(txt)
    else:
        with open(path, "w") as f:
            f.write(txt)


def run_single(ex, ec, sc, x0, tmp, tag, sal, start0=0, want_f64=True):
    """expected -> combine -> score for one saliency, in-process and single-worker, on a temp TSV."""
    tmp = Path(tmp)
    f = tmp / "matrix_chr1.txt"
    write_tsv(f, x0, start0=start0)
    R, N = x0.shape
    out = {}
    if sal == 1:
        cnt = ex.s1Calc(f, Path("null"), (0, R), S, False)
    elif sal == 2:
        cnt = ex.s2Calc(f, Path("null"), (0, R), S, False)
    else:
        cnt = ex.s3Calc(f, (0, R), S, False)
    out["counts"] = cnt
    ex.storeExpArray(cnt, tmp, tag, "matrix_chr1")
    exp_path = tmp / "exp_freq_{}.npy".format(tag)
    ec.main(tmp, exp_path, tag, False)
    q = np.load(exp_path)
    out["exp"] = q
    shared = RawArray(np.ctypeslib.as_ctypes_type(np.float32), R * S)
    sc._init((shared, R, S), exp_path, False)
    if sal == 1:
        sc.s1Score(f, Path("null"), (0, R))
    elif sal == 2:
        sc.s2Score(f, Path("null"), (0, R))
    else:
        sc.s3Score(f, (0, R))
    stored = np.frombuffer(shared, dtype=np.float32).reshape(R, S).copy()
    out["f32"] = stored
    if want_f64 and sal in (1, 2):
        d = x0.astype(int)
        f64 = np.zeros((R, S))
        for r in range(R):
            if sal == 1:
                f64[r] = sc.klScoreND(sc.rowObsS1(d, r, S), q)
            else:
                f64[r] = sc.klScoreND(sc.rowObsS2(d, r, N * (N - 1), S), q).sum(axis=0)
        out["f64"] = f64
    return out, f


def main():
    ex, ec, sc, hp = import_reference()
    print("numpy", np.__version__)
    work = tempfile.mkdtemp(prefix="epg_golden_")

    # ---- real 10-biosample chr1 matrix, built by the reference's own preprocessing script
    gen = Path(work) / "chr1.genome"
    gen.write_text("chr1\t249250621\n")
    subprocess.run(["bash", REF + "/bin/preprocess_data_ChromHMM.sh", REF + "/data/ChromHMM",
                    REF + "/data/metadata_Boix.txt", str(gen)], cwd=work, check=True, stdout=subprocess.DEVNULL)
    full = np.loadtxt(Path(work) / "matrix_chr1.txt", dtype=str, delimiter="\t")
    assert full.shape == (1246253, 13), full.shape
    lo, hi = 50000, 52048
    x_real = (full[lo:hi, 3:].astype(np.int64) - 1).astype(np.int8)
    start0 = int(full[lo, 1])

    real = {"x": x_real, "start0": np.int64(start0)}
    for sal in (1, 2, 3):
        d = tempfile.mkdtemp(dir=work)
        res, f = run_single(ex, ec, sc, x_real, d, "real_s%d" % sal, sal, start0=start0)
        for k, v in res.items():
            real["s%d_%s" % (sal, k)] = v
        if sal == 1:
            # text output exactly as the reference writes it (scores.py:509-536) and countRows (helpers.py:80-99)
            import pandas as pd
            loc = pd.read_table(f, header=None, sep="\t", usecols=[0, 1, 2]).to_numpy()
            outp = Path(d) / "scores.txt.gz"
            sc.writeScores(res["f32"], outp, loc)
            with gzip.open(outp, "rb") as g:
                real["s1_text"] = np.frombuffer(g.read(), dtype=np.uint8)
            real["count_rows"] = np.int64(hp.countRows(f))
    np.savez_compressed(HERE / "real_slice.npz", **real)
    print("real_slice", {k: getattr(v, "shape", None) for k, v in real.items()})

    # ---- synthetic 512 x 833 (S1, S2)
    rng = np.random.default_rng(0)
    x833 = rng.choice(S, size=(512, 833), p=FREQS / FREQS.sum()).astype(np.int8)
    syn = {"x": x833}
    for sal in (1, 2):
        d = tempfile.mkdtemp(dir=work)
        res, _ = run_single(ex, ec, sc, x833, d, "syn_s%d" % sal, sal)
        for k, v in res.items():
            syn["s%d_%s" % (sal, k)] = v
    np.savez_compressed(HERE / "synth833.npz", **syn)
    print("synth833 done")

    # ---- S3 small: N = 32, R = 256
    x32 = rng.choice(S, size=(256, 32), p=FREQS / FREQS.sum()).astype(np.int8)
    d = tempfile.mkdtemp(dir=work)
    res, _ = run_single(ex, ec, sc, x32, d, "s3small", 3)
    np.savez_compressed(HERE / "s3_small.npz", x=x32, **{"s3_" + k: v for k, v in res.items()})
    print("s3_small done")

    # ---- paired 5 + 5 from the real slice
    xa, xb = x_real[:, :5], x_real[:, 5:]
    R = xa.shape[0]
    pair = {"xa": xa, "xb": xb, "qstate": np.int64(S - 1)}
    for sal in (1, 2):
        d = Path(tempfile.mkdtemp(dir=work))
        (d / "A").mkdir(); (d / "B").mkdir()
        fa, fb = d / "A" / "matrix_chr1.txt", d / "B" / "matrix_chr1.txt"
        write_tsv(fa, xa, start0=start0); write_tsv(fb, xb, start0=start0)
        tag = "pair_s%d" % sal
        cnt = (ex.s1Calc if sal == 1 else ex.s2Calc)(fa, fb, (0, R), S, False)
        ex.storeExpArray(cnt, d, tag, "matrix_chr1")
        exp_path = d / "exp_freq_{}.npy".format(tag)
        ec.main(d, exp_path, tag, False)
        arrs = [RawArray(np.ctypeslib.as_ctypes_type(np.float32), R * S) for _ in range(4)]
        quies = RawArray(np.ctypeslib.as_ctypes_type(np.bool_), R)
        sc._initPairwise(arrs[0], arrs[1], arrs[2], arrs[3], quies, R, S, S - 1, exp_path, -1, False)
        # the reference's shuffle is unseeded (helpers.py:183); seed it here and record the uniform draw so the
        # oracle can restate argsort(rand) exactly
        np.random.seed(1234 + sal)
        rand = np.random.rand(R, 10)
        np.random.seed(1234 + sal)
        (sc.s1Score if sal == 1 else sc.s2Score)(fa, fb, (0, R))
        a, b, na, nb = [np.frombuffer(z, dtype=np.float32).reshape(R, S).copy() for z in arrs]
        real_diff = a - b
        null_diff = na - nb
        sign = np.sign(np.sum(null_diff, axis=1))
        null_dist = np.sum(np.square(null_diff), axis=1) * sign
        p = "s%d_" % sal
        pair.update({p + "counts": cnt, p + "exp": np.load(exp_path), p + "a": a, p + "b": b, p + "na": na,
                     p + "nb": nb, p + "delta": real_diff, p + "null_dist": null_dist, p + "rand": rand,
                     p + "quiescent": np.frombuffer(quies, dtype=np.bool_).copy()})
    np.savez_compressed(HERE / "paired.npz", **pair)
    print("paired done")

    # ---- edge cases
    edge = {}
    # (i) state 3 never occurs -> q[3] == 0 for expected; then score a bin that DOES contain state 3 against it
    xe = rng.choice(S, size=(64, 7), p=FREQS / FREQS.sum()).astype(np.int8)
    xe[xe == 3] = 17
    xe[5] = 17            # all-one-state bin
    xe[6] = 0
    d = tempfile.mkdtemp(dir=work)
    res, f = run_single(ex, ec, sc, xe, d, "edge_q0", 1)
    q0 = res["exp"]
    assert q0[3] == 0
    probe = xe.copy(); probe[0, :3] = 3
    p64 = np.stack([sc.klScoreND(sc.rowObsS1(probe.astype(int), r, S), q0) for r in range(probe.shape[0])])
    edge.update(q0_x=xe, q0_exp=q0, q0_counts=res["counts"], q0_probe=probe, q0_probe_f64=p64)
    # S2 with zero cells in q
    d = tempfile.mkdtemp(dir=work)
    res2, _ = run_single(ex, ec, sc, xe, d, "edge_q0_s2", 2)
    edge.update(q0_s2_counts=res2["counts"], q0_s2_exp=res2["exp"], q0_s2_f64=res2["f64"], q0_s2_f32=res2["f32"])
    # (ii) N = 2
    x2 = rng.integers(0, S, size=(128, 2)).astype(np.int8)
    for sal in (1, 2, 3):
        d = tempfile.mkdtemp(dir=work)
        r2, _ = run_single(ex, ec, sc, x2, d, "edge_n2_s%d" % sal, sal)
        for k, v in r2.items():
            edge["n2_s%d_%s" % (sal, k)] = v
    edge["n2_x"] = x2
    # (iii) every state 1..S present in one bin (N = S)
    xs = np.tile(np.arange(S, dtype=np.int8), (4, 1))
    xs[1] = xs[1][::-1]
    d = tempfile.mkdtemp(dir=work)
    r3, _ = run_single(ex, ec, sc, xs, d, "edge_all", 1)
    edge.update(all_x=xs, all_exp=r3["exp"], all_f64=r3["f64"], all_f32=r3["f32"])
    # (iv) file without trailing newline: countRows undercounts by one (helpers.py:94)
    fnt = Path(work) / "nonl.txt"
    write_tsv(fnt, xe[:10], trailing_newline=False)
    edge["nonl_rows"] = np.int64(hp.countRows(fnt))
    fgz = Path(work) / "rows.txt.gz"
    write_tsv(fgz, xe[:10])
    edge["gz_rows"] = np.int64(hp.countRows(fgz))
    edge["split_rows_1246253_8"] = np.array(hp.splitRows(1246253, 8), dtype=np.int64)
    edge["split_rows_7_3"] = np.array(hp.splitRows(7, 3), dtype=np.int64)
    # (v) text formatting of tiny negatives and ordinary values (scores.py:530-531)
    vals = np.array([[-1e-7, 1e-7, 0.0, 3.0769148, -0.36596, 0.49214387, 1.5e-6, -4.9e-6, 2.675, 11.999996] + [0.0] * 8],
                    dtype=np.float32)
    outp = Path(work) / "fmt.txt.gz"
    sc.writeScores(vals, outp, np.array([["chrX", 200, 400]], dtype=object))
    with gzip.open(outp, "rb") as g:
        edge["fmt_text"] = np.frombuffer(g.read(), dtype=np.uint8)
    edge["fmt_vals"] = vals
    np.savez_compressed(HERE / "edge.npz", **edge)
    print("edge done")
    total = sum(os.path.getsize(HERE / n) for n in os.listdir(HERE) if n.endswith(".npz"))
    print("total fixture bytes", total)


if __name__ == "__main__":
    main()

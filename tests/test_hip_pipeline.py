"""GPU end-to-end: the stage drivers and the CLI with the real HIP backend on the reference's golden slice."""
import gzip

import numpy as np
import pytest

from tests.conftest import free_port

torch = pytest.importorskip("torch")

from tests.test_host_logic import write_tsv

pytestmark = pytest.mark.gpu
S = 18


def _text_to_array(txt):
    return np.array([[float(v) for v in l.split("\t")[3:]] for l in txt.decode().splitlines()], dtype=np.float32)


def test_cli_single_s1_real_slice(tmp_path, golden_real):
    from click.testing import CliRunner
    from epilogos_amd import backend
    from epilogos_amd.run import main
    assert backend._override is None           # the product backend, not a stand-in
    g = golden_real
    ind, out = tmp_path / "in10", tmp_path / "out"
    ind.mkdir()
    write_tsv(ind / "matrix_chr1.txt.gz", g["x"], start0=int(g["start0"]))
    meta = tmp_path / "metadata.tsv"
    from tests.conftest import load_golden
    roi = load_golden("roi.npz")
    meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\t%s\n" % (i, i + 1, roi["state_names"][i]) for i in range(S)))
    res = CliRunner().invoke(main, ["-l", "-i", str(ind), "-j", str(meta), "-o", str(out)])
    assert res.exit_code == 0, res.output
    with gzip.open(out / "scores_in10_s1_matrix_chr1.txt.gz", "rb") as fh:
        text = fh.read()
    ref = g["s1_text"].tobytes()
    got_lines, ref_lines = text.split(b"\n"), ref.split(b"\n")
    assert len(got_lines) == len(ref_lines)
    same = sum(a == b for a, b in zip(got_lines, ref_lines))
    print("identical text lines: %d / %d" % (same, len(ref_lines)))
    assert same >= 0.999 * len(ref_lines)
    np.testing.assert_allclose(_text_to_array(text), _text_to_array(ref), atol=1.01e-5)
    # STEP 4 on GPU-produced scores: the reference's regions of interest (scores within 1 float32 ulp of the
    # reference's could reorder exact ties; on this slice the file is identical)
    got = (out / "regionsOfInterest_in10_s1.txt").read_text().splitlines()
    want = roi["roi_single_w50"].tobytes().decode().splitlines()
    assert len(got) == len(want)
    same = sum(a == b for a, b in zip(got, want))
    print("identical ROI lines: %d / %d" % (same, len(want)))
    assert same >= len(want) - 2
    assert not (out / "exp_freq_in10_s1.npy").exists()


@pytest.mark.parametrize("sal", [1, 2])
def test_stage_drivers_hip(tmp_path, golden_real, sal):
    from epilogos_amd import expected, expectedCombination, scores
    g = golden_real
    out = tmp_path / "out"
    out.mkdir()
    f = tmp_path / "matrix_chr1.txt"
    write_tsv(f, g["x"], start0=int(g["start0"]))
    tag = "t_s%d" % sal
    expected.main(f, "null", S, sal, out, tag, 1, False)
    tmp = np.load(out / ("temp_exp_freq_%s_matrix_chr1.npy" % tag))
    assert np.array_equal(tmp, g["s%d_counts" % sal]) and tmp.dtype == np.int64
    expectedCombination.main(out, out / ("exp_freq_%s.npy" % tag), tag, False)
    assert np.array_equal(np.load(out / ("exp_freq_%s.npy" % tag)), g["s%d_exp" % sal])
    scores.main(f, "null", S, sal, out, out / ("exp_freq_%s.npy" % tag), tag, 1, S - 1, -1, False)
    z = np.load(out / ("temp_scores_%s_matrix_chr1.npz" % tag), allow_pickle=True)
    np.testing.assert_allclose(z["scoreArr"], g["s%d_f32" % sal], rtol=3e-7, atol=1e-12)


def test_cli_paired_s1(tmp_path, golden_pair):
    """`epilogos -m paired` with the HIP backend (one rank): STEP 1-3 files through the partitioned driver, then the CLI
    end to end including STEP 4."""
    from click.testing import CliRunner
    from epilogos_amd import driver
    from epilogos_amd.run import main
    from oracle import oracle_np as onp
    from tests.conftest import load_golden
    g = golden_pair
    a, b, out = tmp_path / "male", tmp_path / "female", tmp_path / "out"
    a.mkdir(); b.mkdir(); out.mkdir()
    write_tsv(a / "matrix_chr1.txt.gz", g["xa"]); write_tsv(b / "matrix_chr1.txt.gz", g["xb"])
    meta = tmp_path / "metadata.tsv"
    names = load_golden("roi.npz")["state_names"]
    meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\t%s\n" % (i, i + 1, names[i]) for i in range(S)))
    tag = "male_female_s1"                                           # reference run.py:161-162
    fa, fb = [a / "matrix_chr1.txt.gz"], [b / "matrix_chr1.txt.gz"]
    driver.run_paired_groups(fa, fb, S, 1, out, tag, S - 1, -1, 5)
    assert np.array_equal(np.load(out / ("exp_freq_%s.npy" % tag)), g["s1_exp"])
    with gzip.open(out / ("pairwiseDelta_%s_matrix_chr1.txt.gz" % tag), "rb") as fh:
        delta = _text_to_array(fh.read())
    np.testing.assert_allclose(delta, g["s1_delta"], atol=1.01e-5)
    assert np.array_equal(np.load(out / ("temp_quiescence_%s_matrix_chr1.npz" % tag))["quiescenceArr"], g["s1_quiescent"])
    nd1 = np.load(out / ("temp_nullDistances_%s_matrix_chr1.npz" % tag))["nullDistances"]
    # the side-car of STEP 4 is exactly what the reference would recompute from the text it can read
    side = np.load(out / ("temp_pairMetrics_%s_matrix_chr1.npz" % tag))
    rd, rm = onp.pair_metrics(delta, roundtrip=False)
    assert np.array_equal(side["distances"], rd) and np.array_equal(side["maxDiff"], rm)
    assert np.array_equal(side["starts"], 200 * np.arange(len(rd))) and side["chrName"][0] == "chr1"
    # same seed -> same null draws; quiescent state -1 switches the filter off
    out2 = tmp_path / "out2"
    out2.mkdir()
    driver.run_paired_groups(fa, fb, S, 1, out2, tag, -1, -1, 5)
    assert np.array_equal(nd1, np.load(out2 / ("temp_nullDistances_%s_matrix_chr1.npz" % tag))["nullDistances"])
    assert not np.load(out2 / ("temp_quiescence_%s_matrix_chr1.npz" % tag))["quiescenceArr"].any()

    for flags in ([], ["-n", "-t", "3", "-w", "10"]):
        out3 = tmp_path / ("cli%d" % len(flags))
        res = CliRunner().invoke(main, ["-l", "-m", "paired", "-a", str(a), "-b", str(b), "-j", str(meta), "-o", str(out3),
                                        "--null-seed", "5"] + flags)
        assert res.exit_code == 0, res.output
        assert not list(out3.glob("temp_*.npz")) and not (out3 / ("exp_freq_%s.npy" % tag)).exists()
        with gzip.open(out3 / ("pairwiseMetrics_%s.txt.gz" % tag), "rt") as fh:
            rows = [l.split("\t") for l in fh.read().splitlines()]
        assert len(rows) == len(rd) and len(rows[0]) == (8 if flags else 6)
        got = np.array([float(r[4]) * (1 if r[5] == "+" else -1) for r in rows])
        np.testing.assert_allclose(got, rd, atol=1.01e-5)
        assert [r[3] for r in rows] == [names[m - 1] for m in rm]
        assert (out3 / ("regionsOfInterest_%s.txt" % tag)).exists()
        assert (out3 / ("significantLoci_%s.txt.gz" % tag)).exists() == bool(flags)


@pytest.mark.parametrize("sal", [2, 3])
def test_two_ranks_on_one_gpu_match_one_rank(tmp_path, golden_real, sal):
    """The real HIP backend under torch.distributed.run with two ranks (gloo transport, both on cuda:0): partition,
    device-side count all-reduce and per-rank gzip members give the same files as a single rank."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    from tests.conftest import load_golden
    root = Path(__file__).resolve().parents[1]
    g = golden_real
    ind = tmp_path / "in"
    ind.mkdir()
    write_tsv(ind / "matrix_chr1.txt.gz", g["x"][:1300], chrom="chr1", start0=int(g["start0"]))
    write_tsv(ind / "matrix_chr2.txt.gz", g["x"][1300:], chrom="chr2")
    meta = tmp_path / "metadata.tsv"
    names = load_golden("roi.npz")["state_names"]
    meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\t%s\n" % (i, i + 1, names[i]) for i in range(S)))
    outs = {}
    for world in (1, 2):
        out = tmp_path / ("out%d" % world)
        port = str(free_port())
        env = dict(os.environ, PYTHONPATH=str(root), EPILOGOS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", port, "-m", "epilogos_amd.run", "-l", "-i", str(ind), "-j", str(meta), "-o", str(out),
               "-s", str(sal), "-f", "t"]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
        assert res.returncode == 0, res.stdout + res.stderr
        outs[world] = out
    for name in ("scores_t_matrix_chr1.txt.gz", "scores_t_matrix_chr2.txt.gz"):
        with gzip.open(outs[1] / name, "rb") as a, gzip.open(outs[2] / name, "rb") as b:
            assert a.read() == b.read()
    assert (outs[1] / "regionsOfInterest_t.txt").read_bytes() == (outs[2] / "regionsOfInterest_t.txt").read_bytes()
    with gzip.open(outs[2] / "scores_t_matrix_chr1.txt.gz", "rb") as fh:
        got = _text_to_array(fh.read())
    # and they are the reference's scores (S3: float64 accumulation here against its sequential float32 sum)
    np.testing.assert_allclose(got, g["s%d_f32" % sal][:1300], atol=1.01e-5 if sal == 2 else 2e-5)


def test_driver_uploads_each_part_once_and_rejects_bad_states(tmp_path, golden_real, capsys, monkeypatch):
    """The genome driver on the HIP backend: every part crosses PCIe once (its histograms / matrix stay resident between
    the count pass and the score pass), outputs equal the stage drivers', and a state outside the model stops the run."""
    from epilogos_amd import driver
    g = golden_real
    ind, out = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); out.mkdir()
    write_tsv(ind / "matrix_chr1.txt.gz", g["x"][:1200], start0=int(g["start0"]))
    write_tsv(ind / "matrix_chr2.txt.gz", g["x"][1200:], chrom="chr2")
    files = sorted(ind.glob("*"))
    monkeypatch.setenv("EPILOGOS_TIMING", "1")
    for sal in (1, 2, 3):
        q, results = driver.run_single_group(files, S, sal, out, "t_s%d" % sal)
        assert "H2D uploads: 2 for 2 part(s)" in capsys.readouterr().out
        assert np.array_equal(q, g["s%d_exp" % sal])
        sc = np.concatenate([results["matrix_chr1"][1], results["matrix_chr2"][1]])
        tol = dict(rtol=3e-7, atol=1e-12) if sal < 3 else dict(rtol=1e-4, atol=5e-6)
        np.testing.assert_allclose(sc, g["s%d_f32" % sal], **tol)
    # a file value of 19 in an 18-state model (and 51, which would alias state 19 - 32 in a five-bit decode)
    for bad in (S, S + 32):
        x = g["x"][:64].astype(np.int64).copy()
        x[5, 2] = bad
        bad_dir = tmp_path / ("bad%d" % bad)
        bad_dir.mkdir()
        write_tsv(bad_dir / "matrix_chr1.txt", x)
        with pytest.raises(ValueError):
            driver.run_single_group([bad_dir / "matrix_chr1.txt"], S, 1, out, "bad")
    # a byte the parser cannot see (matrix handed over as an array): the count check after the all-reduce catches it
    from epilogos_amd import backend
    be = backend.HipBackend()
    sess = be.open_single(S, 1)
    xb = g["x"][:100].copy()
    xb[7, 7] = S + 3
    sess.add_part(np.ascontiguousarray(xb), xb.shape[1], 0)
    with pytest.raises(ValueError):
        sess.finish(100, xb.shape[1])


def test_bench_strong_scaling_two_ranks_on_one_gpu(tmp_path):
    """bench.py's multi-rank mode (gloo transport, both ranks on cuda:0): the default is the STRONG split north_star quotes --
    ONE `--bins` genome cut by the splitRows rule -- value counts the genome once, the all-reduced counts cover every bin
    (bench.py asserts sum == bins * biosamples), and --scaling weak holds `--bins` per rank."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    lines = {}
    for mode in ("strong", "weak"):
        port = str(free_port())
        env = dict(os.environ, PYTHONPATH=str(root), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--bins", "300001",
               "--backend", "gloo", "--no-cpu-baseline"] + ([] if mode == "strong" else ["--scaling", "weak"])
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
        assert res.returncode == 0, res.stdout + res.stderr
        lines[mode] = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    s, w = lines["strong"], lines["weak"]
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["config"]["bins_total"] == 300001 and s["config"]["bins_per_gpu"] == 150000
    assert w["scaling"] == "weak" and w["config"]["bins_total"] == 600002 and w["config"]["bins_per_gpu"] == 300001
    for l in (s, w):
        assert abs(l["value"] - l["config"]["bins_total"] / l["ms_per_step"] / 1e3) < 1e-2 * l["value"]
        assert l["roofline"]["bound"] == "hbm" and l["unit"] == "Mbins/s"

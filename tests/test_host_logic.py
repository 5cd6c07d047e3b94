"""CPU tests of the host side: helpers, stage drivers (file names, dtypes, text bytes), partition planning and the
world_size-2 gloo path.  The arithmetic is supplied by an oracle-backed stand-in (tests/fake_backend.py) -- the
product backend needs a GPU and is covered by the -m gpu tests."""
import gzip
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.conftest import free_port

from epilogos_amd import backend, driver, helpers
from tests.fake_backend import OracleBackend

ROOT = Path(__file__).resolve().parents[1]
S = 18


def write_tsv(path, x0, chrom="chr1", start0=0, trailing_newline=True):
    lines = ["{}\t{}\t{}\t{}".format(chrom, start0 + 200 * r, start0 + 200 * r + 200,
                                     "\t".join(str(int(v) + 1) for v in x0[r])) for r in range(x0.shape[0])]
    txt = "\n".join(lines) + ("\n" if trailing_newline else "")
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wt") as f:
        f.write(txt)


@pytest.fixture()
def fake_backend():
    backend.set_for_testing(OracleBackend())
    yield
    backend.set_for_testing(None)


@pytest.fixture()
def state_info(tmp_path):
    p = tmp_path / "metadata.tsv"
    from tests.conftest import load_golden
    names = load_golden("roi.npz")["state_names"]
    p.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\t%s\n" % (i, i + 1, names[i]) for i in range(S)))
    return p


def test_helpers_match_reference_vectors(tmp_path, golden_edge):
    g = golden_edge
    x = g["q0_x"][:10]
    write_tsv(tmp_path / "nonl.txt", x, trailing_newline=False)
    write_tsv(tmp_path / "rows.txt.gz", x)
    assert helpers.countRows(tmp_path / "nonl.txt") == int(g["nonl_rows"]) == 9      # quirk Q6
    assert helpers.countRows(tmp_path / "rows.txt.gz") == int(g["gz_rows"]) == 10
    assert np.array_equal(np.array(helpers.splitRows(1246253, 8)), g["split_rows_1246253_8"])
    assert np.array_equal(np.array(helpers.splitRows(7, 3)), g["split_rows_7_3"])
    got = helpers.readStates(file1Path=tmp_path / "rows.txt.gz", rowsToCalc=(2, 7))
    assert got.dtype == np.int8 and np.array_equal(got, x[2:7])
    assert helpers.strToBool("True") is True and helpers.strToBool("False") is False
    with pytest.raises(ValueError):
        helpers.strToBool("true")
    assert helpers.fileStem("/a/b/matrix_chr1.txt.gz") == "matrix_chr1"


def test_read_states_paired(tmp_path):
    rng = np.random.default_rng(3)
    xa = rng.integers(0, S, size=(20, 4)).astype(np.int8)
    xb = rng.integers(0, S, size=(20, 6)).astype(np.int8)
    write_tsv(tmp_path / "a.txt", xa); write_tsv(tmp_path / "b.txt", xb)
    comb = helpers.readStates(tmp_path / "a.txt", tmp_path / "b.txt", (0, 20), expBool=True)
    assert np.array_equal(comb, np.concatenate([xa, xb], axis=1))
    a, b, sa, sb = helpers.readStates(tmp_path / "a.txt", tmp_path / "b.txt", (0, 20), expBool=False,
                                      rng=np.random.default_rng(0))
    assert sa.shape == (20, 4) and sb.shape == (20, 6)
    # a per-row shuffle keeps every row's multiset of states
    assert np.array_equal(np.sort(np.concatenate([sa, sb], axis=1), axis=1), np.sort(comb, axis=1))
    _, _, sa, sb = helpers.readStates(tmp_path / "a.txt", tmp_path / "b.txt", (0, 20), expBool=False, groupSize=3,
                                      rng=np.random.default_rng(0))
    assert sa.shape == (20, 3) and sb.shape == (20, 3)


@pytest.mark.parametrize("sal", [1, 2, 3])
def test_stage_drivers_reproduce_reference_outputs(tmp_path, golden_real, fake_backend, sal):
    """expected.main -> expectedCombination.main -> scores.main on the real slice: artefact names, dtypes, and for S1
    the exact text the reference wrote (tests/golden/real_slice.npz: s1_text)."""
    from epilogos_amd import expected, expectedCombination, scores
    g = golden_real
    x = g["x"] if sal < 3 else g["x"][:96]
    ind, out = tmp_path / "in10", tmp_path / "out"
    ind.mkdir(); out.mkdir()
    f = ind / "matrix_chr1.txt.gz"
    write_tsv(f, x, start0=int(g["start0"]))
    tag = "in10_s%d" % sal
    expected.main(f, "null", S, sal, out, tag, 1, False)
    tmp = np.load(out / ("temp_exp_freq_%s_matrix_chr1.npy" % tag))
    if sal < 3:
        assert np.array_equal(tmp, g["s%d_counts" % sal]) and tmp.dtype == g["s%d_counts" % sal].dtype
    else:
        assert tmp.dtype == np.int32 and tmp.shape == (10, 10, S, S)
    expPath = out / ("exp_freq_%s.npy" % tag)
    expectedCombination.main(out, expPath, tag, False)
    assert not list(out.glob("temp_exp_freq_*"))                              # temporaries removed
    q = np.load(expPath)
    assert q.dtype == np.float32
    if sal < 3:
        assert np.array_equal(q, g["s%d_exp" % sal])
    scores.main(f, "null", S, sal, out, expPath, tag, 1, S - 1, -1, False)
    with gzip.open(out / ("scores_%s_matrix_chr1.txt.gz" % tag), "rb") as fh:
        text = fh.read()
    z = np.load(out / ("temp_scores_%s_matrix_chr1.npz" % tag), allow_pickle=True)
    assert z["scoreArr"].dtype == np.float32 and z["scoreArr"].shape == (x.shape[0], S)
    assert z["chrName"][0] == "chr1" and z["locationArr"].shape == (x.shape[0], 3)
    if sal == 1:
        assert text == g["s1_text"].tobytes()
        assert np.array_equal(z["scoreArr"], g["s1_f32"])
    if sal == 2:
        np.testing.assert_allclose(z["scoreArr"], g["s2_f32"], rtol=2e-7, atol=1e-12)


def test_paired_stage_drivers(tmp_path, golden_pair, fake_backend):
    from epilogos_amd import expected, expectedCombination, scores
    g = golden_pair
    a, b, out = tmp_path / "A", tmp_path / "B", tmp_path / "out"
    for d in (a, b, out):
        d.mkdir()
    write_tsv(a / "matrix_chr1.txt", g["xa"]); write_tsv(b / "matrix_chr1.txt", g["xb"])
    tag = "A_B_s1"
    expected.main(a / "matrix_chr1.txt", b / "matrix_chr1.txt", S, 1, out, tag, 1, False)
    expectedCombination.main(out, out / ("exp_freq_%s.npy" % tag), tag, False)
    assert np.array_equal(np.load(out / ("exp_freq_%s.npy" % tag)), g["s1_exp"])
    scores.NULL_SEED = 7
    scores.main(a / "matrix_chr1.txt", b / "matrix_chr1.txt", S, 1, out, out / ("exp_freq_%s.npy" % tag), tag, 1,
                int(g["qstate"]), -1, False)
    with gzip.open(out / ("pairwiseDelta_%s_matrix_chr1.txt.gz" % tag), "rt") as fh:
        rows = [l.rstrip("\n").split("\t") for l in fh]
    delta = np.array([[float(v) for v in r[3:]] for r in rows], dtype=np.float32)
    np.testing.assert_allclose(delta, g["s1_delta"], atol=5.1e-6)            # %.5f text
    qz = np.load(out / ("temp_quiescence_%s_matrix_chr1.npz" % tag))
    assert np.array_equal(qz["quiescenceArr"], g["s1_quiescent"])
    nz = np.load(out / ("temp_nullDistances_%s_matrix_chr1.npz" % tag))
    assert nz["nullDistances"].shape == (g["xa"].shape[0],) and nz["nullDistances"].dtype == np.float32


def test_invalid_state_is_reported(tmp_path, fake_backend):
    from epilogos_amd import expected
    x = np.zeros((5, 4), dtype=np.int8)
    x[2, 1] = S          # file value S+1: outside the model
    write_tsv(tmp_path / "m.txt", x)
    with pytest.raises(ValueError):
        expected.main(tmp_path / "m.txt", "null", S, 1, tmp_path, "t", 1, False)
    with pytest.raises(ValueError):
        expected.main(tmp_path / "m.txt", "null", S, 4, tmp_path, "t", 1, False)


def test_plan_partition_properties():
    rows = [1246253, 7, 0, 999, 51304566 // 200]
    total = sum(rows)
    for world in (1, 2, 3, 8, 16):
        plans = driver.plan_partition(rows, world)
        assert len(plans) == world
        covered = np.zeros(total, dtype=np.int8) if total < 5_000_000 else None
        seen = {f: [] for f in range(len(rows))}
        for parts in plans:
            for (f, lo, hi) in parts:
                assert 0 <= lo < hi <= rows[f]
                seen[f].append((lo, hi))
        for f, ivs in seen.items():            # each file's rows are covered exactly once, in order
            ivs.sort()
            pos = 0
            for lo, hi in ivs:
                assert lo == pos
                pos = hi
            assert pos == rows[f]
        sizes = [sum(hi - lo for _, lo, hi in p) for p in plans]
        assert max(sizes) - min(sizes) <= 1     # splitRows balance


def _decompressed(path):
    with gzip.open(path, "rb") as fh:
        return fh.read()


def _io_passes(log):
    """{input path: [(operation, lo, hi)]} from an EPILOGOS_IO_LOG file: every entry is one pass over the whole file."""
    passes = {}
    for line in Path(log).read_text().splitlines():
        _pid, op, path, lo, hi = line.split("\t")
        passes.setdefault(path, []).append((op, int(lo), int(hi)))
    return passes


def test_two_rank_gloo_matches_single_process(tmp_path, golden_real):
    """world_size 2 over gloo on CPU: partition + all-reduce + per-rank gzip members == the single-process output."""
    g = golden_real
    ind = tmp_path / "in"
    ind.mkdir()
    x = g["x"]
    write_tsv(ind / "matrix_chr1.txt", x[:1500], chrom="chr1", start0=int(g["start0"]))
    write_tsv(ind / "matrix_chr2.txt", x[1500:], chrom="chr2")
    outs = {}
    for world in (1, 2):
        out = tmp_path / ("out%d" % world)
        out.mkdir()
        env = dict(os.environ, PYTHONPATH=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500),
                   EPILOGOS_IO_LOG=str(tmp_path / ("io%d.log" % world)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"],
               str(ROOT / "tests" / "gloo_worker.py"), str(ind), str(out)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        outs[world] = out
        # every input file is inflated and parsed exactly ONCE in total, whole, whatever the number of ranks (round 2: every
        # rank made a counting pass over every file before the first byte was parsed)
        assert _io_passes(tmp_path / ("io%d.log" % world)) == {str(ind / "matrix_chr1.txt"): [("read", 0, -1)],
                                                               str(ind / "matrix_chr2.txt"): [("read", 0, -1)]}
    for name in ("scores_t_s1_matrix_chr1.txt.gz", "scores_t_s1_matrix_chr2.txt.gz"):
        assert _decompressed(outs[1] / name) == _decompressed(outs[2] / name)
    assert np.array_equal(np.load(outs[1] / "exp_freq_t_s1.npy"), np.load(outs[2] / "exp_freq_t_s1.npy"))
    assert np.array_equal(np.load(outs[1] / "exp_freq_t_s1.npy"), g["s1_exp"])     # == the reference's exp_freq
    z1 = np.load(outs[1] / "temp_scores_t_s1_matrix_chr2.npz", allow_pickle=True)
    z2 = np.load(outs[2] / "temp_scores_t_s1_matrix_chr2.npz", allow_pickle=True)
    assert np.array_equal(z1["scoreArr"], z2["scoreArr"]) and z2["chrName"][0] == "chr2"
    # chr1 holds the first 1500 bins of the slice: its text is the reference's first 1500 lines
    ref_lines = g["s1_text"].tobytes().split(b"\n")
    assert _decompressed(outs[2] / "scores_t_s1_matrix_chr1.txt.gz") == b"\n".join(ref_lines[:1500]) + b"\n"


def test_two_rank_gloo_paired_matches_single_process(tmp_path, golden_pair):
    g = golden_pair
    ind = tmp_path / "in"
    (ind / "A").mkdir(parents=True); (ind / "B").mkdir()
    for name, lo, hi in (("matrix_chr1.txt", 0, 1100), ("matrix_chr2.txt", 1100, 2048)):
        write_tsv(ind / "A" / name, g["xa"][lo:hi], chrom=name[7:-4])
        write_tsv(ind / "B" / name, g["xb"][lo:hi], chrom=name[7:-4])
    outs = {}
    for world in (1, 2):
        out = tmp_path / ("out%d" % world)
        out.mkdir()
        port = str(free_port())
        env = dict(os.environ, PYTHONPATH=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, EPILOGOS_IO_LOG=str(tmp_path / ("io%d.log" % world)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", port, str(ROOT / "tests" / "gloo_worker.py"), str(ind), str(out)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        outs[world] = out
        passes = _io_passes(tmp_path / ("io%d.log" % world))
        assert len(passes) == 4 and all(v == [("read", 0, -1)] for v in passes.values()), passes
    for stem in ("matrix_chr1", "matrix_chr2"):
        assert _decompressed(outs[1] / ("pairwiseDelta_t_s1_%s.txt.gz" % stem)) == _decompressed(outs[2] / ("pairwiseDelta_t_s1_%s.txt.gz" % stem))
        for kind, key in (("temp_nullDistances", "nullDistances"), ("temp_quiescence", "quiescenceArr")):
            a = np.load(outs[1] / ("%s_t_s1_%s.npz" % (kind, stem)))
            b = np.load(outs[2] / ("%s_t_s1_%s.npz" % (kind, stem)))
            assert np.array_equal(a[key], b[key]) and a["chrName"][0] == b["chrName"][0] == stem[7:]
    assert np.array_equal(np.load(outs[2] / "exp_freq_t_s1.npy"), g["s1_exp"])
    q = np.concatenate([np.load(outs[2] / ("temp_quiescence_t_s1_%s.npz" % s))["quiescenceArr"] for s in ("matrix_chr1", "matrix_chr2")])
    assert np.array_equal(q, g["s1_quiescent"])


def test_cli_single_mode(tmp_path, golden_real, fake_backend, state_info):
    from click.testing import CliRunner
    from epilogos_amd.run import main
    g = golden_real
    ind, out = tmp_path / "in10", tmp_path / "out"
    ind.mkdir()
    write_tsv(ind / "matrix_chr1.txt.gz", g["x"], start0=int(g["start0"]))
    res = CliRunner().invoke(main, ["-l", "-i", str(ind), "-j", str(state_info), "-o", str(out), "-c", "8"])
    assert res.exit_code == 0, res.output
    assert _decompressed(out / "scores_in10_s1_matrix_chr1.txt.gz") == g["s1_text"].tobytes()   # default tag: {dir}_s{sal}
    # a second run through the binary input cache gives the same bytes
    out2 = tmp_path / "out_cached"
    for _ in range(2):
        res = CliRunner().invoke(main, ["-l", "-i", str(ind), "-j", str(state_info), "-o", str(out2), "--cache-dir", str(tmp_path / "cache")])
        assert res.exit_code == 0, res.output
        assert _decompressed(out2 / "scores_in10_s1_matrix_chr1.txt.gz") == g["s1_text"].tobytes()
    assert len(list((tmp_path / "cache").glob("matrix_chr1_*.npy"))) == 4       # states, location blob + offsets, state range
    os.environ.pop("EPILOGOS_CACHE_DIR", None)
    # STEP 4 ran: the reference's regions of interest, and its clean-up of exp_freq / temp_scores (quirk Q4)
    from tests.conftest import load_golden
    assert (out / "regionsOfInterest_in10_s1.txt").read_bytes() == load_golden("roi.npz")["roi_single_w50"].tobytes()
    assert not (out / "exp_freq_in10_s1.npy").exists() and not list(out.glob("temp_scores_*"))
    res = CliRunner().invoke(main, ["-i", str(ind), "-j", str(state_info), "-o", str(out), "-s", "3", "-m", "paired",
                                    "-a", str(ind), "-b", str(ind)])
    assert res.exit_code == 0 and "ERROR" in res.output                                     # -i with paired mode


def test_cli_paired_mode_to_step4(tmp_path, golden_pair, fake_backend, state_info):
    """`epilogos -m paired` end to end: STEP 1-3 through the partitioned driver, then STEP 4.  With the reference's
    arithmetic behind the backend the final files are the reference's own, byte for byte (z-score branch and, with -n,
    the p-value branch; vectors of tests/golden/make_golden_pairwise.py)."""
    from click.testing import CliRunner
    from epilogos_amd.run import main
    from tests.conftest import load_golden
    g, g4 = golden_pair, load_golden("pairwise_step4.npz")
    a, b = tmp_path / "male", tmp_path / "female"
    a.mkdir(); b.mkdir()
    start0 = int(g4["start0"])
    for name, (lo, hi) in zip(g4["split_names"], g4["split_bounds"]):
        for d, x in ((a, g["xa"]), (b, g["xb"])):
            write_tsv(d / "matrix_{}.txt.gz".format(name), x[lo:hi], chrom=str(name), start0=start0 + 200 * int(lo))
    tag = "male_female_s1"                                           # reference run.py:161-162
    for flags, kind in (([], "z"), (["-n", "-t", "3"], "p")):
        out = tmp_path / ("out_" + kind)
        res = CliRunner().invoke(main, ["-l", "-m", "paired", "-a", str(a), "-b", str(b), "-j", str(state_info), "-o", str(out),
                                        "--null-seed", "5"] + flags)
        assert res.exit_code == 0, res.output
        assert _decompressed(out / "pairwiseDelta_{}_matrix_chr10.txt.gz".format(tag)).count(b"\n") == 600
        assert not list(out.glob("temp_*.npz")) and not (out / "exp_freq_{}.npy".format(tag)).exists()
        if kind == "z":                                              # deterministic: no null distances involved
            assert _decompressed(out / "pairwiseMetrics_{}.txt.gz".format(tag)) == g4["real_metrics_z"].tobytes()
            assert (out / "regionsOfInterest_{}.txt".format(tag)).read_bytes() == g4["real_roi_z_w125"].tobytes()
        else:                                                        # the null shuffle differs from the reference's draw
            lines = _decompressed(out / "pairwiseMetrics_{}.txt.gz".format(tag)).split(b"\n")[:-1]
            ref = g4["real_metrics_z"].tobytes().split(b"\n")[:-1]
            assert len(lines) == 2048 and all(l.count(b"\t") == 7 and l.startswith(r) for l, r in zip(lines, ref))
            assert (out / "significantLoci_{}.txt.gz".format(tag)).exists() and (out / "regionsOfInterest_{}.txt".format(tag)).exists()
    r = CliRunner().invoke(main, ["-l", "-m", "paired", "-a", str(a), "-b", str(b), "-j", str(state_info), "-o", str(tmp_path / "o3"),
                                  "-t", "0"])
    assert "Number of trials" in r.output


def test_cli_argument_errors(tmp_path, state_info, fake_backend):
    """Error behaviour of the reference's checkFlags / checkArguments (run.py:328-451) for the STEP 1-3 flags."""
    from click.testing import CliRunner
    from epilogos_amd.run import main
    ind = tmp_path / "in"
    ind.mkdir()
    write_tsv(ind / "m.txt", np.zeros((4, 3), dtype=np.int8))
    run = lambda *a: CliRunner().invoke(main, list(a))
    assert "required" in run("-l", "-o", str(tmp_path / "o"), "-j", str(state_info)).output          # no -i
    assert "required" in run("-l", "-i", str(ind), "-j", str(state_info)).output                      # no -o
    assert "required" in run("-l", "-i", str(ind), "-o", str(tmp_path / "o")).output                   # no -j
    r = run("-l", "-i", str(ind), "-o", str(tmp_path / "o"), "-j", str(state_info), "-s", "4")
    assert isinstance(r.exception, ValueError)
    r = run("-l", "-m", "paired", "-a", str(ind), "-b", str(ind), "-o", str(tmp_path / "o"), "-j", str(state_info), "-s", "3")
    assert isinstance(r.exception, ValueError)                                                          # paired supports S1/S2
    r = run("-l", "-i", str(tmp_path / "missing"), "-o", str(tmp_path / "o"), "-j", str(state_info))
    assert isinstance(r.exception, FileNotFoundError)
    empty = tmp_path / "empty"
    empty.mkdir()
    r = run("-l", "-i", str(empty), "-o", str(tmp_path / "o"), "-j", str(state_info))
    assert isinstance(r.exception, OSError)
    assert "same as the input" in run("-l", "-i", str(ind), "-o", str(ind), "-j", str(state_info)).output
    assert "Version" in run("-v").output
    b = tmp_path / "b"
    b.mkdir()
    write_tsv(b / "other.txt", np.zeros((4, 3), dtype=np.int8))
    r = run("-l", "-m", "paired", "-a", str(ind), "-b", str(b), "-o", str(tmp_path / "o2"), "-j", str(state_info))
    assert isinstance(r.exception, FileNotFoundError)                                                   # no same-named file in -b


def test_stem_that_prefixes_another_stem(tmp_path, golden_real, fake_backend):
    """hg19 has chr1 next to chr1_gl000191_random: a file's parts are named from the partition plan (file index + first
    row), never rediscovered by a prefix glob; part files left by a crashed run with the same tag are removed."""
    x = golden_real["x"]
    ind, out = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); out.mkdir()
    write_tsv(ind / "m_chr1.txt.gz", x[:30], chrom="chr1")
    write_tsv(ind / "m_chr1_gl000191_random.txt.gz", x[30:50], chrom="chr1_gl000191_random")
    (out / ".part_scores_t_s1_f0000_000000000007.gz").write_bytes(b"stale")        # leftover of an earlier crash
    files = sorted(ind.glob("*"))
    _, results = driver.run_single_group(files, S, 1, out, "t_s1")
    a = _decompressed(out / "scores_t_s1_m_chr1.txt.gz").splitlines()
    b = _decompressed(out / "scores_t_s1_m_chr1_gl000191_random.txt.gz").splitlines()
    assert len(a) == 30 and len(b) == 20
    assert all(l.startswith(b"chr1\t") for l in a) and all(l.startswith(b"chr1_gl000191_random\t") for l in b)
    assert results["m_chr1"][1].shape == (30, S) and results["m_chr1_gl000191_random"][1].shape == (20, S)
    assert not list(out.glob(".part_*"))


def test_two_rank_gloo_prefix_stems(tmp_path, golden_real):
    """The same layout over two ranks: the range border cuts the first file, its gzip members are concatenated by exact
    name, and the arrays for STEP 4 reach rank 0 through send/recv."""
    x = golden_real["x"]
    ind = tmp_path / "in"
    ind.mkdir()
    write_tsv(ind / "m_chr1.txt", x[:700], chrom="chr1")
    write_tsv(ind / "m_chr1_gl000191_random.txt", x[700:1000], chrom="chr1_gl000191_random")
    outs = {}
    for world in (1, 2):
        out = tmp_path / ("out%d" % world)
        out.mkdir()
        port = str(free_port())
        env = dict(os.environ, PYTHONPATH=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", port, str(ROOT / "tests" / "gloo_worker.py"), str(ind), str(out)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        outs[world] = out
    for stem, n in (("m_chr1", 700), ("m_chr1_gl000191_random", 300)):
        one, two = _decompressed(outs[1] / ("scores_t_s1_%s.txt.gz" % stem)), _decompressed(outs[2] / ("scores_t_s1_%s.txt.gz" % stem))
        assert one == two and len(one.splitlines()) == n
        z1 = np.load(outs[1] / ("temp_scores_t_s1_%s.npz" % stem), allow_pickle=True)
        z2 = np.load(outs[2] / ("temp_scores_t_s1_%s.npz" % stem), allow_pickle=True)
        assert np.array_equal(z1["scoreArr"], z2["scoreArr"]) and np.array_equal(z1["locationArr"], z2["locationArr"])
    assert not list(outs[2].glob(".part_*"))


def test_three_rank_gloo_uneven_files_text_then_cache(tmp_path, golden_real):
    """Four files of very different sizes over three ranks: the long second file is parsed by the middle rank, whose bin
    range covers only its middle -- the head goes to rank 0, the tail to rank 2 (driver._redistribute); outputs equal the
    single-process ones byte for byte and every file is read once.  A second run with the --cache-dir side-cars takes the
    "ranges" route: no pass over any text file, each rank memory-maps its own row ranges."""
    x = golden_real["x"]
    ind = tmp_path / "in"
    ind.mkdir()
    cuts = [0, 90, 1400, 1460, x.shape[0]]
    for k in range(4):
        write_tsv(ind / ("m_chr%d.txt" % (k + 1)), x[cuts[k]:cuts[k + 1]], chrom="chr%d" % (k + 1))
    cache = tmp_path / "cache"
    outs = {}
    for tag, world, use_cache in (("one", 1, False), ("three", 3, False), ("fill", 1, True), ("cached", 3, True)):
        out = tmp_path / ("out_" + tag)
        out.mkdir()
        port = str(free_port())
        env = dict(os.environ, PYTHONPATH=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, EPILOGOS_IO_LOG=str(tmp_path / (tag + ".log")))
        env.pop("EPILOGOS_CACHE_DIR", None)
        if use_cache:
            env["EPILOGOS_CACHE_DIR"] = str(cache)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", port, str(ROOT / "tests" / "gloo_worker.py"), str(ind), str(out)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        outs[tag] = out
    for tag in ("one", "three", "fill"):
        passes = _io_passes(tmp_path / (tag + ".log"))
        assert sorted(passes) == sorted(str(ind / ("m_chr%d.txt" % k)) for k in (1, 2, 3, 4)), (tag, passes)
        assert all(v == [("read", 0, -1)] for v in passes.values()), (tag, passes)
    assert not (tmp_path / "cached.log").exists()                # nothing was inflated or parsed: the cache served row ranges
    for k in (1, 2, 3, 4):
        name = "scores_t_s1_m_chr%d.txt.gz" % k
        want = _decompressed(outs["one"] / name)
        assert len(want.splitlines()) == cuts[k] - cuts[k - 1]
        for tag in ("three", "fill", "cached"):
            assert _decompressed(outs[tag] / name) == want, (tag, name)
        z1 = np.load(outs["one"] / ("temp_scores_t_s1_m_chr%d.npz" % k), allow_pickle=True)
        for tag in ("three", "cached"):
            z = np.load(outs[tag] / ("temp_scores_t_s1_m_chr%d.npz" % k), allow_pickle=True)
            assert np.array_equal(z1["scoreArr"], z["scoreArr"]) and np.array_equal(z1["locationArr"], z["locationArr"])
    for tag in ("three", "cached"):
        assert np.array_equal(np.load(outs["one"] / "exp_freq_t_s1.npy"), np.load(outs[tag] / "exp_freq_t_s1.npy"))
        assert not list(outs[tag].glob(".part_*"))


def test_more_ranks_than_files_and_an_empty_file(tmp_path, golden_real):
    """Four ranks, three files of which one is empty: two ranks parse nothing and own bins all the same (everything they score
    was handed over), the empty file still gets its (empty) output, outputs equal the single-process ones."""
    x = golden_real["x"]
    ind = tmp_path / "in"
    ind.mkdir()
    write_tsv(ind / "m_chr1.txt", x[:1200], chrom="chr1")
    (ind / "m_chr2.txt").write_text("")
    write_tsv(ind / "m_chr3.txt", x[1200:], chrom="chr3")
    outs = {}
    for world in (1, 4):
        out = tmp_path / ("out%d" % world)
        out.mkdir()
        port = str(free_port())
        env = dict(os.environ, PYTHONPATH=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, EPILOGOS_IO_LOG=str(tmp_path / ("io%d.log" % world)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", port, str(ROOT / "tests" / "gloo_worker.py"), str(ind), str(out)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        outs[world] = out
        passes = _io_passes(tmp_path / ("io%d.log" % world))
        assert len(passes) == 3 and all(v == [("read", 0, -1)] for v in passes.values()), passes
    for stem, n in (("m_chr1", 1200), ("m_chr2", 0), ("m_chr3", x.shape[0] - 1200)):
        one, four = _decompressed(outs[1] / ("scores_t_s1_%s.txt.gz" % stem)), _decompressed(outs[4] / ("scores_t_s1_%s.txt.gz" % stem))
        assert one == four and len(one.splitlines()) == n
    assert np.array_equal(np.load(outs[1] / "exp_freq_t_s1.npy"), np.load(outs[4] / "exp_freq_t_s1.npy"))
    assert not list(outs[4].glob(".part_*"))


def test_cli_rejects_states_outside_the_model(tmp_path, golden_real, fake_backend, state_info):
    """The reference dies with an IndexError when the data holds a state the -j model does not have (expected.py:113); the
    command line here must not run to completion either -- neither for a value just above the model nor for one that would
    alias a valid state in the kernels' five-bit decode (18 + 32)."""
    from click.testing import CliRunner
    from epilogos_amd.run import main
    for bad in (S, S + 32, -1):                       # 0-based: file values 19, 51, 0
        ind, out = tmp_path / ("in%d" % bad), tmp_path / ("out%d" % bad)
        ind.mkdir()
        x = golden_real["x"][:40].astype(np.int64).copy()
        x[17, 3] = bad
        write_tsv(ind / "matrix_chr1.txt", x)
        res = CliRunner().invoke(main, ["-l", "-i", str(ind), "-j", str(state_info), "-o", str(out)])
        assert res.exit_code != 0
        assert isinstance(res.exception, ValueError) and "outside the 18-state model" in str(res.exception)
        assert not list(out.glob("scores_*"))


def test_gpus_launcher_command(tmp_path, state_info, capsys, monkeypatch):
    """`--gpus N` (reference: the SLURM fan-out of run.py:190-279): the child command is torch.distributed.run with one process
    per GPU on 127.0.0.1 and this module as its program, --gpus itself stripped; a process already under the launcher
    (WORLD_SIZE set) never launches again."""
    from epilogos_amd import run
    argv = ["-l", "-i", "in", "--gpus", "4", "-j", "meta.tsv", "-o", "out", "--gpus=4", "-s", "2"]
    assert run._strip_gpus(argv) == ["-l", "-i", "in", "-j", "meta.tsv", "-o", "out", "-s", "2"]
    cmd = run._launch_command(4, argv, 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    k = cmd.index("epilogos_amd.run")
    assert cmd[k - 1] == "-m" and cmd[k + 1:] == run._strip_gpus(argv)
    # dry run through the real option parser: the launch happens after the arguments have been checked, before torch is imported
    ind = tmp_path / "in"
    ind.mkdir()
    (ind / "m_chr1.txt").write_text("chr1\t0\t200\t1\t2\n")
    monkeypatch.setenv("EPILOGOS_LAUNCH_DRYRUN", "1")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = ["-l", "-i", str(ind), "-j", str(state_info), "-o", str(tmp_path / "out"), "--gpus", "3"]
    monkeypatch.setattr(run, "_ARGV", args)
    with pytest.raises(SystemExit) as e:
        run.main(args=args, standalone_mode=False)
    assert e.value.code == 0
    printed = capsys.readouterr().out
    assert "torch.distributed.run" in printed and "--nproc-per-node 3" in printed and "--gpus" not in printed
    with pytest.raises(SystemExit):
        run.main(args=args[:-1] + ["-2"], standalone_mode=False)


def test_visible_gpus_is_counted_without_torch(monkeypatch):
    """`--gpus 0` (all visible GPUs): the launcher parent counts them from the KFD topology / the *_VISIBLE_DEVICES lists, not
    through torch or HIP -- it must stay GPU-free, its children are the ranks."""
    import subprocess
    from epilogos_amd import run
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    base = run._visible_gpus()
    assert base >= 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert run._visible_gpus() == (3 if base == 1 else min(base, 3))
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0")
    assert run._visible_gpus() == 1
    code = "import sys; from epilogos_amd import run; run._visible_gpus(); assert 'torch' not in sys.modules"
    assert subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=str(ROOT))).returncode == 0


def test_early_readers_do_not_outlive_a_failed_start(tmp_path, golden_real, state_info):
    """The single-process command line starts its file readers before it imports torch (round 4).  When the stage driver never
    takes them over -- here: no GPU, the product backend raises -- they are aborted: the process ends at once with an error
    instead of waiting for readers that wait for a session.  (With the oracle stand-in installed the same readers feed the run:
    test_cli_single_mode and test_cli_paired_mode_to_step4 go through them.)"""
    import time
    ind = tmp_path / "in"
    ind.mkdir()
    for k in range(3):
        write_tsv(ind / ("m_chr%d.txt.gz" % (k + 1)), golden_real["x"][k * 300:(k + 1) * 300], chrom="chr%d" % (k + 1))
    env = dict(os.environ, PYTHONPATH=str(ROOT), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    res = subprocess.run([sys.executable, "-m", "epilogos_amd.run", "-l", "-i", str(ind), "-j", str(state_info), "-o", str(tmp_path / "out")],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "no HIP device" in res.stderr or "EpilogosHipError" in res.stderr or "HIP" in res.stderr, res.stderr[-500:]
    assert time.time() - t0 < 120
    assert not list((tmp_path / "out").glob("scores_*"))


def test_get_num_states_counts_whitespace_only_lines_like_pandas(tmp_path):
    """pandas' read_table (reference helpers.py:9-17) skips empty lines and lines of spaces; a line of tabs (the
    separator) is a row."""
    import pandas as pd
    from epilogos_amd import helpers
    for k, text in enumerate(("a\tb\n1\t2\n\n3\t4\n", "a\tb\n1\t2\n\t\n3\t4\n", "a\tb\n1\t2\n \n3\t4\n\n\n", "a\tb\r\n1\t2\r\n\r\n3\t4\r\n",
                              "a\tb\n1\t2\n \t \n3\t4\n", "a\tb\n1\t2\n   \n\t\n")):
        f = tmp_path / ("m%d.tsv" % k)
        f.write_text(text)
        assert helpers.getNumStates(f) == pd.read_table(f, header=0, sep="\t").shape[0], repr(text)


def test_placement_decision_on_faked_ratio_sequences(monkeypatch):
    """engine.alloc_hist's searches on the host with the probe faked.  The first job on a matrix gets a plain allocation and no
    search, the second the QUICK search (eight blocks, 50 ms), the fourth -- if the plain allocation was kept -- the DEEP one
    (24 blocks), once.  Candidate 0 is the PLAIN allocation the search replaces; the decision is RELATIVE (two levels of the ratio
    >= 3 % apart: the lower one is another memory class) and settled by ONE comparison over the whole matrix against the plain
    allocation -- the placed cache is never slower than what it replaces.  Sequences: a first block that is clearly good; a run
    of the matrix's own class with a good block behind it; a walk that only ever sees one level (bounded, the plain allocation
    stays, the deep search finds the block behind it); the driver's round-5 box (a block that passes its slices at 1.105 and
    loses over the whole matrix, 1.174: the matrix straddles there); a plain allocation that already lies in another class;
    +-1 % of clock wobble inside a level."""
    import torch
    from epilogos_amd import engine
    order = {}

    def level_of(t):
        return order.setdefault(t.untyped_storage().data_ptr(), len(order))

    def run(levels, whole=None, jobs=None, R=4096, S=18, ldx=848):
        """levels[k]: slice ratio of the k-th distinct candidate the searches touch (0 = the first plain allocation); whole[k]: its
        time over the whole matrix (default: ratio + 0.02).  -> the report after `jobs` jobs on the matrix."""
        order.clear()
        jobs = jobs or engine.PLACE_SEARCH_AT
        X = torch.zeros((R, ldx), dtype=torch.int8)
        keep = []                                                         # (the host allocator must not hand a freed block out again)

        def fake_probe(X_, N_, S_, Hflat, counts, slices, reps=2):
            if Hflat is None:
                return 1.0
            keep.append(Hflat)
            k = level_of(Hflat)
            over_whole = slices == [(0, R)]
            return (whole or {}).get(k, levels[min(k, len(levels) - 1)] + 0.02) if over_whole else levels[min(k, len(levels) - 1)]

        monkeypatch.setattr(engine, "_probe_ms", fake_probe)
        monkeypatch.setattr(engine, "PLACE_MIN_BYTES", 1024)
        monkeypatch.setattr(engine, "PLACE_BLOCK", 1 << 16)
        monkeypatch.setattr(engine, "placement_enabled", lambda: True)
        monkeypatch.setattr(engine, "_order_after_last_user", lambda st: None)
        monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: None)
        monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
        monkeypatch.setattr(engine, "_probe_slices", lambda R_, rows=0: [(0, 1024), (1024, 2048), (3072, 4096)])
        monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (1 << 40, 1 << 40))
        engine.release_placement()
        reps = []
        for j in range(jobs):
            H = engine.alloc_hist(X, ldx - 15, S)
            assert H.shape == (R, S) and H.dtype == torch.int16
            reps.append(engine.placement_report(X.device) if X.device.index is not None else dict(engine._placement[None]["report"]))
            del H
        engine.release_placement()
        assert reps[0] == {"jobs_seen": 1, "tier": "none yet"}             # the first job never searches
        rep = reps[-1]
        if "blocks_tried" in rep:
            assert rep["left_in_torch_cache_GiB"] >= 0 and "search_ms" in rep and rep["good"] == (rep["picked"] != 0)
        return rep

    rep = run([1.17, 1.08])                                               # the first block is clearly another class: no walk
    assert rep["tier"] == "quick" and rep["good"] and rep["decision"] == "sure" and rep["blocks_tried"] == 1 and rep["picked"] == 1
    rep = run([1.17, 1.08], jobs=6)                                       # ... and the later jobs reuse the home, no second search
    assert rep["tier"] == "quick" and rep["reuses"] == 4 and rep["jobs_seen"] == 2
    rep = run([1.175, 1.17, 1.18, 1.13, 1.17])                            # 3.5 % under the run before it: two levels, no absolute level involved
    assert rep["good"] and rep["decision"] == "two-levels" and rep["blocks_tried"] == 3 and rep["picked"] == 3
    one_level = [1.17, 1.165, 1.175, 1.168, 1.172, 1.17, 1.166, 1.174, 1.169]     # one level, +-0.5 % of wobble ...
    rep = run(one_level + [1.17] * 12 + [1.08])
    assert not rep["good"] and rep["blocks_tried"] == engine.PLACE_TRIES and rep["picked"] == 0       # ... bounded at eight blocks: the plain allocation stays
    assert rep["decision"].startswith("plain allocation kept (one-level")
    rep = run(one_level + [1.17] * 12 + [1.08], jobs=3)                   # the third job: nothing new
    assert rep["tier"] == "quick" and rep["jobs_seen"] == 3
    rep = run(one_level + [1.17] * 12 + [1.08], jobs=engine.PLACE_DEEP_AT + 2)    # the fourth: the deep search, once, walks past the run
    assert rep["tier"] == "deep" and rep["good"] and rep["decision"] == "sure" and 8 < rep["blocks_tried"] <= engine.PLACE_DEEP_TRIES
    assert rep["quick"]["blocks_tried"] == engine.PLACE_TRIES and rep["reuses"] == 2
    rep = run([1.17] * 60, jobs=engine.PLACE_DEEP_AT + 3)                 # no good block anywhere: both searches bounded, then never again
    assert rep["tier"] == "deep" and not rep["good"] and rep["blocks_tried"] == engine.PLACE_DEEP_TRIES and rep["jobs_seen"] == engine.PLACE_DEEP_AT + 3
    rep = run([1.14, 1.105, 1.10], whole={0: 1.17, 1: 1.174, 2: 1.12})    # the driver's box of round 5: block 1 passes its slices ...
    assert rep["good"] and rep["picked"] == 2 and rep["lost_over_the_whole_matrix"] == [1]         # ... loses over the whole matrix, the walk goes on, block 2 wins
    rep = run([1.14, 1.09] + [1.145] * 8, whole={0: 1.17, 1: 1.19})       # sure on the slices, slower than the plain allocation over the whole matrix;
    assert not rep["good"] and rep["picked"] == 0 and rep["lost_over_the_whole_matrix"][0] == 1 and rep["blocks_tried"] == 8   # nothing better behind it
    rep = run([1.09, 1.17])                                               # the plain allocation already lies in another class: kept, no block tried
    assert not rep["good"] and rep["blocks_tried"] == 0 and rep["decision"] == "plain allocation kept (sure)"
    rep = run([1.16, 1.16, 1.12] + [1.17] * 8, whole={0: 1.18, 2: 1.21})  # two levels, the lower one loses over the whole matrix: plain stays
    assert not rep["good"] and rep["lost_over_the_whole_matrix"][0] == 2 and rep["picked"] == 0
    rep = run([1.16, 1.16, 1.12] + [1.17] * 8, whole={0: 1.18, 2: 1.175}) # ... wins by less than 1 %: not worth a home block
    assert not rep["good"] and rep["picked"] == 0
    # the decision function alone: wobble inside a level never reads as two levels, a real step does, excluded picks are skipped
    assert engine.place_decide([1.14, 1.15, 1.145, 1.16, 1.136])[1] == "one-level"
    assert engine.place_decide([1.14, 1.15, 1.105]) == (2, "two-levels")
    assert engine.place_decide([1.14, 1.10, 1.15]) == (1, "sure")
    assert engine.place_decide([1.17, 1.12, 1.17, 1.118], excluded=[3]) == (1, "two-levels")
    assert engine.place_decide([], ()) == (None, "none")


def test_placement_is_off_for_ranks_sharing_a_device(monkeypatch):
    import torch
    from epilogos_amd import engine
    if getattr(torch._C, "_storage_Use_Count", None) is None:
        pytest.skip("this torch cannot count storage users: placement is off altogether")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("EPILOGOS_PLACEMENT", raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert engine.placement_enabled()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert not engine.placement_enabled()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    monkeypatch.setenv("EPILOGOS_PLACEMENT", "0")
    assert not engine.placement_enabled()

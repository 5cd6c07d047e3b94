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
    # round 3: the command line gathers S1 scores from a table built with the reference's own numpy expression
    # (scores.s1ScoreTable), so the text IS the reference's, byte for byte (round 2: >= 99.9 % of the lines)
    assert text == ref
    # STEP 4 on GPU-produced scores: the reference's regions of interest, the same file
    got = (out / "regionsOfInterest_in10_s1.txt").read_text().splitlines()
    want = roi["roi_single_w50"].tobytes().decode().splitlines()
    assert got == want
    assert not (out / "exp_freq_in10_s1.npy").exists()


def test_cli_single_s1_on_input_only_pandas_reads(tmp_path, golden_real, capsys):
    """The golden slice written the way pandas (the reference's parser, helpers.py:152-155) reads it and the strict native parser
    does not -- a blank line in the middle, "+7", " 7", "7.0" -- next to a clean second file: the command line re-reads the one
    file through pandas (into the staging buffer the native attempt had already taken), says so, and writes the reference's text for
    both files."""
    from click.testing import CliRunner
    from epilogos_amd.run import main
    g = golden_real
    ind, out = tmp_path / "in10", tmp_path / "out"
    ind.mkdir()
    write_tsv(ind / "matrix_chr1.txt", g["x"], start0=int(g["start0"]))
    lines = (ind / "matrix_chr1.txt").read_text().split("\n")
    assert lines[-1] == ""
    for r, form in ((3, "+{}"), (700, " {}"), (1500, "{}.0")):
        f = lines[r].split("\t")
        f[5] = form.format(f[5])
        lines[r] = "\t".join(f)
    lines.insert(1000, "")                                             # a blank line inside the file
    (ind / "matrix_chr1.txt").write_text("\n".join(lines))
    write_tsv(ind / "matrix_chr2.txt", g["x"][:500], chrom="chr2")
    meta = tmp_path / "metadata.tsv"
    from tests.conftest import load_golden
    names = load_golden("roi.npz")["state_names"]
    meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\t%s\n" % (i, i + 1, names[i]) for i in range(S)))
    res = CliRunner().invoke(main, ["-l", "-i", str(ind), "-j", str(meta), "-o", str(out)])
    assert res.exit_code == 0, res.output
    assert "through pandas" in res.output and res.output.count("through pandas") == 1
    # the same two files, clean: identical outputs
    ind2, out2 = tmp_path / "clean" / "in10", tmp_path / "out2"
    ind2.mkdir(parents=True)
    write_tsv(ind2 / "matrix_chr1.txt", g["x"], start0=int(g["start0"]))
    write_tsv(ind2 / "matrix_chr2.txt", g["x"][:500], chrom="chr2")
    res2 = CliRunner().invoke(main, ["-l", "-i", str(ind2), "-j", str(meta), "-o", str(out2)])
    assert res2.exit_code == 0 and "through pandas" not in res2.output
    for name in ("scores_in10_s1_matrix_chr1.txt.gz", "scores_in10_s1_matrix_chr2.txt.gz"):
        with gzip.open(out / name, "rb") as a, gzip.open(out2 / name, "rb") as b:
            ta, tb = a.read(), b.read()
        assert ta == tb and ta.count(b"\n") == (2048 if "chr1" in name else 500)
    assert (out / "regionsOfInterest_in10_s1.txt").read_text() == (out2 / "regionsOfInterest_in10_s1.txt").read_text()


def test_single_session_counts_and_scores_parts_in_batches(monkeypatch):
    """S1 parts of less than a GiB are counted in ONE launch per batch (epg_bin_hist_parts, histograms in one flat allocation with
    unwritten rows between the parts) and scored in ONE launch per batch: the same counts, exp_freq and float32 scores, bit for bit,
    as a launch per part (EPILOGOS_SINGLE_BATCH=0) and as the oracle -- with row counts that are no multiple of eight, an empty
    part, a one-row part, parts asked for out of order, a part sliced and one dropped before the batch was counted, and a batch
    that is only partly scored early."""
    from epilogos_amd import backend
    from oracle import oracle_np as onp
    from tests.conftest import synth_states
    N = 70
    rows = [1001, 0, 1, 4096, 777, 13]
    xs = [synth_states(r, N, seed=40 + k) for k, r in enumerate(rows)]
    allx = np.concatenate([x for x in xs if len(x)])
    q_ref = onp.normalise(onp.expected_s1(allx, S))
    want = [onp.score_s1(x, q_ref, S).astype(np.float32) if len(x) else np.zeros((0, S), np.float32) for x in xs]
    be = backend.HipBackend()

    def job(batch, early):
        monkeypatch.setenv("EPILOGOS_SINGLE_BATCH", "1" if batch else "0")
        sess = be.open_single(S, 1)
        pids = [sess.add_device(be.to_device(x) if len(x) else torch.empty((0, 80), dtype=torch.int8, device="cuda"), N if len(x) else 0) for x in xs]
        if batch:
            assert sum(p is backend._PENDING for p in sess.parts) == 5 and sess._pending_rows == sum(rows)
        extra = sess.slice_part(pids[3], 100, 1100)                       # (forces the count pass of the pending batch)
        assert not sess._pending and all(p is not backend._PENDING for p in sess.parts)
        sess.ensure_acc(N)
        total = sum(rows)
        sess.launch(total, N, [pids[k] for k in early] + [extra])
        q = sess.finish(total, N)
        out = {k: sess.scores(pids[k]) for k in (5, 0, 4, 3, 2, 1)}      # out of order; some were scored by launch(), some are now
        return q, out, sess.scores(extra)

    for early in ([0, 1, 2, 3, 4, 5], [0, 2], []):
        qa, a, ea = job(True, early)
        qb, b, eb = job(False, early)
        assert np.array_equal(qa, q_ref) and np.array_equal(qb, q_ref)
        for k in range(len(rows)):
            assert a[k].shape == (rows[k], S) and np.array_equal(a[k], b[k]), (k, early)
            np.testing.assert_allclose(a[k], want[k], rtol=2e-7, atol=0)
        assert np.array_equal(ea, a[3][100:1100]) and np.array_equal(eb, ea)


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
    if sal == 1:
        assert np.array_equal(z["scoreArr"], g["s1_f32"])       # host-built table: the reference's float32 values, bit for bit


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


def _cli_inputs(tmp_path, g, S_=S):
    from tests.conftest import load_golden
    ind = tmp_path / "in"
    ind.mkdir()
    write_tsv(ind / "matrix_chr1.txt.gz", g["x"][:1300], chrom="chr1", start0=int(g["start0"]))
    write_tsv(ind / "matrix_chr2.txt.gz", g["x"][1300:], chrom="chr2")
    meta = tmp_path / "metadata.tsv"
    names = load_golden("roi.npz")["state_names"]
    meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\t%s\n" % (i, i + 1, names[i]) for i in range(S_)))
    return ind, meta


def test_cli_one_rank_group_over_rccl(tmp_path, golden_real):
    """The command line under torch.distributed.run with ONE rank and the default backend (nccl = RCCL): the process group,
    _Dist.comm_device, the device-side all-reduce of the count tensor (S1: int64[18]; S3: int32[N, N, S, S]) and
    destroy_process_group run on RCCL; the files equal those of a plain start."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    ind, meta = _cli_inputs(tmp_path, golden_real)
    for sal in (1, 3):
        outs = {}
        for how in ("plain", "rccl"):
            out = tmp_path / ("out_%s_%d" % (how, sal))
            port = str(free_port())
            env = dict(os.environ, PYTHONPATH=str(root), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, EPILOGOS_TIMING="1")
            env.pop("EPILOGOS_DIST_BACKEND", None)
            tail = ["-m", "epilogos_amd.run", "-l", "-i", str(ind), "-j", str(meta), "-o", str(out), "-s", str(sal), "-f", "t"]
            cmd = [sys.executable] + (tail if how == "plain" else
                                      ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                                       "--master-port", port] + tail)
            res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
            assert res.returncode == 0, res.stdout + res.stderr
            outs[how] = out
        for name in ("scores_t_matrix_chr1.txt.gz", "scores_t_matrix_chr2.txt.gz"):
            with gzip.open(outs["plain"] / name, "rb") as a, gzip.open(outs["rccl"] / name, "rb") as b:
                assert a.read() == b.read()
        assert (outs["plain"] / "regionsOfInterest_t.txt").read_bytes() == (outs["rccl"] / "regionsOfInterest_t.txt").read_bytes()


def test_s3_count_tensor_of_833_biosamples_through_rccl(tmp_path):
    """The 899 MB int32[833, 833, 18, 18] count tensor of BASELINE config 4 goes through dist.all_reduce on RCCL (a group of
    one rank: what this box can hold) inside the command line's session, and the scores still equal the oracle's."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    code = """
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from epilogos_amd import backend, driver
from tests.conftest import synth_states
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
d = driver._Dist()
assert d.comm_device is not None and d.world == 1
S, N, R = 18, 833, 600
x = synth_states(R, N, seed=11)
be = backend.HipBackend()
sess = be.open_single(S, 3)
pid = sess.add_device(be.to_device(x), N)
assert sess.acc.numel() == N * N * S * S and sess.acc.dtype == torch.int32
before = sess.acc.clone()
sess.all_reduce(d)                                   # RCCL, 899 MB, in place
assert torch.equal(before, sess.acc)
q = sess.finish(R, N)
sc = sess.scores(pid)
from oracle import oracle_np as onp
qo = onp.normalise(onp.expected_s3(x, S))
assert np.array_equal(q, qo)
ref = onp.score_s3_f64(x[:24], qo, S)
np.testing.assert_allclose(sc[:24], ref, rtol=1e-6, atol=1e-6)
print("S3_RCCL_OK", float(sc.sum()))
dist.destroy_process_group()
""" % str(root)
    port = str(free_port())
    env = dict(os.environ, PYTHONPATH=str(root), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert res.returncode == 0 and "S3_RCCL_OK" in res.stdout, res.stdout + res.stderr


def test_gpus_flag_starts_one_process_per_rank(tmp_path, golden_real):
    """`--gpus 2` from a plain start: the command line itself starts torch.distributed.run as a child (before it touches a
    GPU) and returns its exit code.  On a one-GPU box the two ranks share cuda:0 over gloo (EPILOGOS_DIST_BACKEND); the
    text inputs take the parse-once route (each file on one rank, border pieces handed over) and the outputs equal the
    single-process ones."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    ind, meta = _cli_inputs(tmp_path, golden_real)
    outs = {}
    for gpus in (1, 2):
        out = tmp_path / ("out%d" % gpus)
        env = dict(os.environ, PYTHONPATH=str(root), EPILOGOS_DIST_BACKEND="gloo", EPILOGOS_IO_LOG=str(tmp_path / ("io%d.log" % gpus)))
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, "-m", "epilogos_amd.run", "-l", "-i", str(ind), "-j", str(meta), "-o", str(out), "-s", "1", "-f", "t",
               "--gpus", str(gpus)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
        assert res.returncode == 0, res.stdout + res.stderr
        assert ("GPUs = %d" % gpus) in res.stdout
        outs[gpus] = out
        reads = [l.split("\t") for l in (tmp_path / ("io%d.log" % gpus)).read_text().splitlines()]
        assert sorted(r[2] for r in reads) == sorted(str(p) for p in ind.glob("*"))     # each file once, whatever the rank count
    for name in ("scores_t_matrix_chr1.txt.gz", "scores_t_matrix_chr2.txt.gz"):
        with gzip.open(outs[1] / name, "rb") as a, gzip.open(outs[2] / name, "rb") as b:
            assert a.read() == b.read()
    assert (outs[1] / "regionsOfInterest_t.txt").read_bytes() == (outs[2] / "regionsOfInterest_t.txt").read_bytes()
    # a failing child's exit code is the parent's
    bad = subprocess.run([sys.executable, "-m", "epilogos_amd.run", "-l", "-i", str(tmp_path / "nowhere"), "-j", str(meta), "-o",
                          str(tmp_path / "o"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=600, cwd=str(root))
    assert bad.returncode != 0


@pytest.mark.parametrize("sal", [1, 2])
def test_paired_two_ranks_on_one_gpu_match_one_rank(tmp_path, golden_pair, sal):
    """Paired mode over two ranks (gloo transport, both on cuda:0), text inputs: each of the four files is parsed by one rank,
    the border pieces' histograms change hands (backend._HipPairedSession.export_rows / import_rows), the null shuffle is keyed
    by the global bin index -- pairwiseDelta, the STEP 4 files and exp_freq equal the single-rank run's."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    from tests.conftest import load_golden
    root = Path(__file__).resolve().parents[1]
    g = golden_pair
    a, b = tmp_path / "male", tmp_path / "female"
    a.mkdir(); b.mkdir()
    for name, lo, hi in (("matrix_chr1.txt.gz", 0, 1300), ("matrix_chr2.txt.gz", 1300, 2048)):
        write_tsv(a / name, g["xa"][lo:hi], chrom=name[7:-7])
        write_tsv(b / name, g["xb"][lo:hi], chrom=name[7:-7])
    meta = tmp_path / "metadata.tsv"
    names = load_golden("roi.npz")["state_names"]
    meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\t%s\n" % (i, i + 1, names[i]) for i in range(S)))
    outs = {}
    for gpus in (1, 2):
        out = tmp_path / ("out%d" % gpus)
        env = dict(os.environ, PYTHONPATH=str(root), EPILOGOS_DIST_BACKEND="gloo", EPILOGOS_IO_LOG=str(tmp_path / ("io%d.log" % gpus)))
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, "-m", "epilogos_amd.run", "-l", "-m", "paired", "-a", str(a), "-b", str(b), "-j", str(meta), "-o", str(out),
               "-s", str(sal), "-f", "t", "--null-seed", "77", "-w", "10", "--gpus", str(gpus)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
        assert res.returncode == 0, res.stdout + res.stderr
        outs[gpus] = out
        reads = sorted(l.split("\t")[2] for l in (tmp_path / ("io%d.log" % gpus)).read_text().splitlines())
        assert reads == sorted(str(p) for d in (a, b) for p in d.glob("*"))                 # four files, each parsed once
    assert np.array_equal(np.load(outs[1] / "exp_freq_t.npy"), np.load(outs[2] / "exp_freq_t.npy")) if (outs[1] / "exp_freq_t.npy").exists() else True
    for name in ("pairwiseDelta_t_matrix_chr1.txt.gz", "pairwiseDelta_t_matrix_chr2.txt.gz", "pairwiseMetrics_t.txt.gz"):
        with gzip.open(outs[1] / name, "rb") as fa, gzip.open(outs[2] / name, "rb") as fb:
            assert fa.read() == fb.read(), name
    assert (outs[1] / "regionsOfInterest_t.txt").read_bytes() == (outs[2] / "regionsOfInterest_t.txt").read_bytes()
    if sal == 1:
        with gzip.open(outs[2] / "pairwiseDelta_t_matrix_chr1.txt.gz", "rb") as fh:
            np.testing.assert_allclose(_text_to_array(fh.read()), g["s1_delta"][:1300], atol=1.01e-5)


def test_bench_one_rank_group_and_graph_replay(tmp_path):
    """bench.py --pg --graph on one GPU: the step (K1, RCCL all-reduce of a one-rank group, combine, score) is captured in a
    hipGraph and replayed; the line reports the all-reduce by itself and says how the step was launched."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PYTHONPATH=str(root))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, str(root / "bench.py"), "--steps", "5", "--warmup", "2", "--bins", "400000", "--no-cpu-baseline", "--configs", "none"]
    lines = {}
    for tag, extra in (("plain", []), ("pg", ["--pg"]), ("graph", ["--pg", "--graph"])):
        res = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
        assert res.returncode == 0, res.stdout + res.stderr
        lines[tag] = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert lines["plain"]["allreduce_probe"] is None
    for tag in ("pg", "graph"):
        p = lines[tag]["allreduce_probe"]
        assert p["world"] == 1 and p["backend"] == "nccl" and p["device_us_per_call_back_to_back"] > 0 and p["host_us_per_call"] > 0
    assert lines["graph"]["config"]["step_launch"] == "hipGraph replay"
    for l in lines.values():
        assert l["kernels_ms"]["k_bin_hist"] > 0 and l["value"] > 0


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


def test_bench_gpus_flag_from_a_plain_start(tmp_path):
    """`python bench.py --gpus 2 ...` with no launcher around it: bench.py starts its own ranks (child torchrun, before it
    touches a GPU), rank 0's one JSON line arrives on the parent's stdout and carries what every rank saw."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PYTHONPATH=str(root))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--bins", "300001", "--backend", "gloo",
           "--configs", "none"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout + res.stderr
    out = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["bins_per_gpu"] == 150000
    assert line["cpu_baseline"] is None                              # N = 1 only
    pr = line["per_rank"]
    assert len(pr["k_bin_hist_ms"]) == 2 and pr["k_bin_hist_ms_min_max"][0] <= pr["k_bin_hist_ms_min_max"][1]
    assert pr["skew_ms_per_step"] >= 0 and len(pr["own_ms_per_step"]) == 2
    assert line["allreduce_probe"]["world"] == 2
    assert line["config"]["step_path"].startswith("backend._HipSingleSession: add_device(X, N) -> all_reduce -> launch") and line["s1_table"]["built_on"].startswith("device")
    # one GPU: unchanged contract, plus both S1 paths on the genome and on the shard
    one = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "3", "--warmup", "1", "--bins", "400000", "--shard-bins", "50000",
                          "--no-cpu-baseline", "--configs", "none", "--placement-experiment", "0"], env=env, capture_output=True, text=True,
                         timeout=900, cwd=str(tmp_path))
    assert one.returncode == 0, one.stdout + one.stderr
    l1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert l1["n_gpus"] == 1 and l1["per_rank"] is None
    assert set(l1["s1_paths"]) >= {"genome_400000_bins", "shard_50000_bins"}
    for k in ("genome_400000_bins", "shard_50000_bins"):
        assert l1["s1_paths"][k]["session_ms_per_step"] > 0 and l1["s1_paths"][k]["engine_ms_per_step"] > 0


@pytest.mark.parametrize("mode", ["single", "paired"])
def test_owner_slice_that_starts_mid_file_at_an_odd_row(tmp_path, golden_real, golden_pair, mode):
    """Two ranks, files of 101 + 1000 rows: rank 1 parses the second file itself and owns its rows from 449 on -- a row slice
    of the resident histograms whose base is 449 * 18 * 2 bytes = 4 mod 8 past an aligned allocation.  The score entry points
    want aligned bases: the slice is copied when it is not.  Outputs equal the one-rank run."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    from tests.conftest import load_golden
    root = Path(__file__).resolve().parents[1]
    meta = tmp_path / "metadata.tsv"
    names = load_golden("roi.npz")["state_names"]
    meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\t%s\n" % (i, i + 1, names[i]) for i in range(S)))
    cuts = (("matrix_chr1.txt.gz", 0, 101), ("matrix_chr2.txt.gz", 101, 1101))
    if mode == "single":
        ind = tmp_path / "in"
        ind.mkdir()
        for name, lo, hi in cuts:
            write_tsv(ind / name, golden_real["x"][lo:hi], chrom=name[7:-7])
        inputs = ["-i", str(ind)]
        outputs = ["scores_t_matrix_chr1.txt.gz", "scores_t_matrix_chr2.txt.gz"]
    else:
        a, b = tmp_path / "male", tmp_path / "female"
        a.mkdir(); b.mkdir()
        for name, lo, hi in cuts:
            write_tsv(a / name, golden_pair["xa"][lo:hi], chrom=name[7:-7])
            write_tsv(b / name, golden_pair["xb"][lo:hi], chrom=name[7:-7])
        inputs = ["-m", "paired", "-a", str(a), "-b", str(b), "--null-seed", "3", "-w", "10"]
        outputs = ["pairwiseDelta_t_matrix_chr1.txt.gz", "pairwiseDelta_t_matrix_chr2.txt.gz", "pairwiseMetrics_t.txt.gz"]
    outs = {}
    for gpus in (1, 2):
        out = tmp_path / ("out%d" % gpus)
        env = dict(os.environ, PYTHONPATH=str(root), EPILOGOS_DIST_BACKEND="gloo")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, "-m", "epilogos_amd.run", "-l"] + inputs + ["-j", str(meta), "-o", str(out), "-s", "1", "-f", "t", "--gpus", str(gpus)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
        assert res.returncode == 0, res.stdout + res.stderr
        outs[gpus] = out
    for name in outputs:
        with gzip.open(outs[1] / name, "rb") as fa, gzip.open(outs[2] / name, "rb") as fb:
            assert fa.read() == fb.read(), name


def test_paired_driver_15_states_odd_last_tile(tmp_path):
    """Paired S1 through the driver for a 15-state model whose file ends on an odd number of bins in its last 64-bin tile (two
    files: 317 and 63 bins): every delta -- the LAST state of the LAST bin included, which the one-pass kernel's loader got from
    stale LDS until round 4 -- equals the oracle's, and the quiescence mask and STEP 4's side-car are the oracle's."""
    from epilogos_amd import driver
    from oracle import oracle_np as onp
    from tests.conftest import synth_states
    S_, NA, NB = 15, 9, 7
    a, b, out = tmp_path / "A", tmp_path / "B", tmp_path / "out"
    a.mkdir(); b.mkdir(); out.mkdir()
    xs = {}
    for name, R, seed in (("m_chr1.txt.gz", 317, 1), ("m_chr2.txt.gz", 63, 2)):
        xa, xb = synth_states(R, NA, S=S_, seed=seed), synth_states(R, NB, S=S_, seed=seed + 10)
        xa[3:6, :] = S_ - 1
        xb[3:5, :] = S_ - 1
        write_tsv(a / name, xa, chrom=name[2:-7]); write_tsv(b / name, xb, chrom=name[2:-7])
        xs[name] = (xa, xb)
    fa, fb = sorted(a.glob("*")), sorted(b.glob("*"))
    for rep in range(2):                                             # (stale LDS differs from run to run)
        driver.run_paired_groups(fa, fb, S_, 1, out, "t", S_ - 1, -1, 11)
        cat = np.concatenate([np.concatenate(xs[n], axis=1) for n in sorted(xs)])
        q = onp.normalise(onp.expected_s1(cat, S_))
        assert np.array_equal(np.load(out / "exp_freq_t.npy"), q)
        for name in sorted(xs):
            xa, xb = xs[name]
            want, _ = onp.pair_finish(onp.score_s1(xa, q, S_).astype(np.float32), onp.score_s1(xb, q, S_).astype(np.float32))
            with gzip.open(out / ("pairwiseDelta_t_%s.txt.gz" % name[:-7]), "rb") as fh:
                got = _text_to_array(fh.read())
            np.testing.assert_allclose(got, want, atol=1.01e-5)
            assert abs(got[-1, -1] - want[-1, -1]) <= 1.01e-5
            qm = np.load(out / ("temp_quiescence_t_%s.npz" % name[:-7]))["quiescenceArr"]
            assert np.array_equal(qm, onp.quiescent_mask(xa, xb, S_ - 1)) and qm.sum() == 2
            side = np.load(out / ("temp_pairMetrics_t_%s.npz" % name[:-7]))
            rd, rm = onp.pair_metrics(got, roundtrip=False)
            assert np.array_equal(side["distances"], rd) and np.array_equal(side["maxDiff"], rm)


def test_bench_eight_ranks_on_one_gpu(tmp_path):
    """bench.py's N = 8 code path with all its configs (S2, S3, paired in chromosome parts, per-rank gather, all-reduce probe),
    eight gloo ranks sharing cuda:0, small sizes: the numbers mean nothing (eight processes on one GPU), the line must be complete --
    the first real 8-GPU run is the driver's, this is what can be checked before."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PYTHONPATH=str(root))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "8", "--backend", "gloo", "--bins", "800008", "--s3-bins", "80000",
           "--s3-small-bins", "40000", "--steps", "3", "--warmup", "1", "--config-reps", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout + res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["bins_per_gpu"] == 100001
    assert len(line["per_rank"]["k_bin_hist_ms"]) == 8 and line["allreduce_probe"]["world"] == 8
    for name in ("s2", "s3", "s3_small", "paired"):
        cfg = line["configs"][name]
        assert "error" not in cfg and cfg["job_ms"] > 0 and cfg["bins_total"] in (800008, 80000, 40000), cfg
    assert line["configs"]["paired"]["outputs_finite"] and line["configs"]["s2"]["scores_finite"]
    assert line["cpu_baseline"] is None and line["s1_paths"] is None
    # round 5: the run explains itself -- who drove which device, the step as one graph replay, every collective checked
    who = line["per_rank"]["ranks"]
    assert len(who) == 8 and sorted(w["local_rank"] for w in who) == list(range(8))
    assert all(w["hip_device"] == 0 and "name" in w and "cores_allowed" in w for w in who)
    assert line["graph_ms_per_step"] is None or line["graph_ms_per_step"] > 0     # (a host backend cannot be captured: error reported)
    assert ("graph_error" in line) == (line["graph_ms_per_step"] is None)
    st = line["rccl_selftest"]
    assert st["ok"] and st["ok_on_every_rank"] and st["world"] == 8 and st["backend"] == "gloo"
    assert sum(1 for v in st.values() if isinstance(v, dict) and v.get("ok")) == 6


def test_bench_extras_timeout_exits_nonzero_after_the_line(tmp_path):
    """A secondary measurement that never returns (here: forced in the S2 leg of two gloo ranks on cuda:0) must not look like a
    clean run: rank 0 prints the headline with a `deadline` note, then every rank leaves with exit code 3 and the launcher
    reports failure."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PYTHONPATH=str(root), EPG_BENCH_HANG_LEG="s2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--bins", "200000", "--steps", "2",
           "--warmup", "1", "--config-reps", "1", "--extras-deadline", "25", "--graph-leg", "0"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode != 0, res.stdout[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "exit code 3" in line["configs"]["deadline"] and "'s2'" in line["configs"]["deadline"]
    assert "s2" not in line["configs"]
    # one process, no launcher: the exit code is bench.py's own
    env1 = dict(env, EPG_BENCH_HANG_LEG="unplaced")
    res1 = subprocess.run([sys.executable, str(root / "bench.py"), "--bins", "1500000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                           "--extras-deadline", "20", "--graph-leg", "0", "--shard-bins", "0"], env=env1, capture_output=True, text=True,
                          timeout=600, cwd=str(tmp_path))
    assert res1.returncode == 3, res1.stdout[-2000:] + res1.stderr[-2000:]
    assert "exit code 3" in json.loads([l for l in res1.stdout.splitlines() if l.startswith("{")][-1])["configs"]["deadline"]


def test_bench_line_survives_rank0_dying_in_a_secondary_measurement(tmp_path):
    """More than one rank: the line is printed by bench.LineKeeper, a child of rank 0 that holds its latest version.  Rank 0 is
    aborted (SIGABRT, what a failed assertion inside a collective library does) in the paired leg of two gloo ranks: the launcher
    reports failure and the ONE line on stdout is the state before that leg -- headline and the S2 config in it, an
    `ended_early` note, no paired config."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, PYTHONPATH=str(root), EPG_BENCH_ABORT_LEG="paired")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--bins", "200000", "--steps", "2",
           "--warmup", "1", "--config-reps", "1", "--configs", "s2,paired", "--extras-deadline", "120", "--graph-leg", "0"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode != 0, res.stdout[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert "'paired'" in line["ended_early"]
    assert "s2" in line["configs"] and "paired" not in line["configs"]
    # and the undisturbed run of the same command prints one complete line through the same keeper
    env.pop("EPG_BENCH_ABORT_LEG")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert "ended_early" not in line and "paired" in line["configs"] and "s2" in line["configs"]


@pytest.mark.parametrize("mode", ["single", "paired"])
def test_launch_then_finish_rescores_when_a_device_table_differs(mode):
    """The one sequence of the command line and of bench.py: launch() (STEP 2 on the device + every part's score pass enqueued, no
    host sync), then finish() (count check, device-built S1 tables against numpy's).  The branch nobody has ever seen taken -- a
    device table that differs from the reference's -- is forced here by corrupting the table between the two calls: finish() must
    replace it AND score the early parts again, so that what scores() / results() hand out are the reference's numbers."""
    import torch
    from epilogos_amd import backend
    from epilogos_amd.driver import shuffle_key
    from oracle import oracle_np as onp
    from tests.conftest import synth_states
    be = backend.HipBackend()
    S_, NA, NB, R_ = 18, 61, 47, 500
    xa, xb = synth_states(R_, NA, seed=5), synth_states(R_, NB, seed=6)

    def run(corrupt):
        if mode == "single":
            sess = be.open_single(S_, 1)
            pids = [sess.add_device(be.to_device(xa[:300]), NA), sess.add_device(be.to_device(xa[300:]), NA)]
            sess.ensure_acc(NA)
            sess.launch(R_, NA, pids)
            early = [sess.early_scores(p).clone() for p in pids]
        else:
            sess = be.open_paired(S_, 1, S_ - 1, -1, 99)
            pids = [sess.add_staged(be.to_device(xa[:300]), NA, be.to_device(xb[:300]), NB, shuffle_key(0, 0)),
                    sess.add_staged(be.to_device(xa[300:]), NA, be.to_device(xb[300:]), NB, shuffle_key(1, 0))]
            sess.ensure_acc(NA + NB)
            sess.launch(R_, NA + NB, pids)
            early = [sess._early[p]["delta"].clone() for p in pids]
        if corrupt:
            for T in sess._t1.values():
                T.view(-1)[T.numel() // 2] += 1.0                  # one float32 entry of every device-built table
        q = sess.finish(R_, NA if mode == "single" else NA + NB)
        out = [sess.scores(p) if mode == "single" else sess.results(p)["delta"] for p in pids]
        return q, np.concatenate(out), torch.cat(early).cpu().numpy(), sess.tables_patched

    q0, clean, early0, patched0 = run(False)
    q1, fixed, early1, patched1 = run(True)
    assert patched0 == 0 and patched1 >= 1
    assert np.array_equal(q0, q1) and np.array_equal(clean, early0)      # nothing patched: the early results ARE the results
    assert np.array_equal(fixed, clean)                                   # patched: re-scored from the reference's table
    if mode == "single":
        ref = onp.score_s1(xa, q0, S_).astype(np.float32)
        np.testing.assert_allclose(clean, ref, rtol=2e-7, atol=0)
    else:
        both = np.concatenate([xa, xb], axis=1)
        assert np.array_equal(q0, onp.normalise(onp.expected_s1(both, S_)))
        ref = onp.score_s1(xa, q0, S_).astype(np.float32) - onp.score_s1(xb, q0, S_).astype(np.float32)
        np.testing.assert_allclose(clean, ref, rtol=1e-6, atol=1e-7)

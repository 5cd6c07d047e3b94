"""What only several ranks on one node would hit: every rank keeps to its share of the host's cores, a
rank that fails takes the job down at once instead of leaving the others in the all-reduce, files that are all cut by range
borders change hands completely, and a cache that only some ranks can see does not split the ranks over two protocols."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from tests.conftest import ROOT, free_port
from tests.test_host_logic import _decompressed, write_tsv


def _run(ind, out, world, env_extra=None, timeout=600):
    out.mkdir(exist_ok=True)
    port = str(free_port())
    env = dict(os.environ, PYTHONPATH=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    for k in ("EPILOGOS_CACHE_DIR", "EPILOGOS_NUM_CORES", "EPILOGOS_HOST_THREADS", "EPILOGOS_PARSE_WORKERS", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", port, str(ROOT / "tests" / "gloo_worker.py"), str(ind), str(out)]
    t0 = time.time()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    return res, time.time() - t0


def test_host_budget_arithmetic(monkeypatch):
    from epilogos_amd import _io
    node = _io.node_cores()
    for k in ("EPILOGOS_NUM_CORES", "LOCAL_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert _io.host_budget() == node and os.environ["EPILOGOS_HOST_THREADS"] == str(node)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert _io.host_budget() == max(1, node // 4)
    monkeypatch.setenv("EPILOGOS_NUM_CORES", "2")                     # the reference's -c as an upper bound for the whole job
    assert _io.host_budget() == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert _io.host_budget() == min(2, node)
    assert _io.load().epgio_default_threads() == min(2, node)           # what `threads=0` means inside the native library
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1000")
    assert _io.host_budget() == 1                                     # never less than one


@pytest.mark.parametrize("cap", [0, 4])
def test_four_ranks_share_the_hosts_cores(tmp_path, golden_real, cap):
    """Four ranks, twelve gzip files: every rank sizes its parser pool, its per-file native threads and its writers from
    node cores / LOCAL_WORLD_SIZE (and from -c when given), so the ranks' peaks of runnable library threads add up to no more
    than the node has (round 3: every rank took the whole node -- 8 x 3 x 5 threads on a 16-core quota)."""
    from epilogos_amd import _io
    x = golden_real["x"]
    ind = tmp_path / "in"
    ind.mkdir()
    for k in range(12):
        write_tsv(ind / ("m_chr%02d.txt.gz" % (k + 1)), x[k * 170:(k + 1) * 170], chrom="chr%d" % (k + 1))
    log = tmp_path / "threads.log"
    extra = {"EPILOGOS_THREAD_LOG": str(log)}
    if cap:
        extra["EPILOGOS_NUM_CORES"] = str(cap)
    res, _ = _run(ind, tmp_path / "out4", 4, extra)
    assert res.returncode == 0, res.stdout + res.stderr
    rows = [l.split("\t") for l in log.read_text().splitlines()]
    assert len(rows) == 4 and len({r[0] for r in rows}) == 4          # one line per rank
    node = min(_io.node_cores(), cap) if cap else _io.node_cores()
    share = max(1, node // 4)
    assert all(int(r[2]) == share and int(r[4]) == 4 for r in rows), rows
    assert all(1 <= int(r[1]) <= share for r in rows), rows          # a rank's peak stays inside its share ...
    assert sum(int(r[1]) for r in rows) <= max(node, 4)               # ... so the node's budget holds
    one, _ = _run(ind, tmp_path / "out1", 1)
    assert one.returncode == 0, one.stdout + one.stderr
    for k in range(12):
        name = "scores_t_s1_m_chr%02d.txt.gz" % (k + 1)
        assert _decompressed(tmp_path / "out4" / name) == _decompressed(tmp_path / "out1" / name)


def test_a_failing_rank_ends_the_job_quickly(tmp_path, golden_real):
    """Three ranks; the file the LAST rank parses carries a state outside the model.  That rank raises before the count
    all-reduce; the job must come back non-zero within seconds -- not sit in the collective until a timeout."""
    x = golden_real["x"]
    ind = tmp_path / "in"
    ind.mkdir()
    for k in range(3):
        part = x[k * 600:(k + 1) * 600].astype(np.int64).copy()
        if k == 2:
            part[17, 3] = 18                                        # file value 19 in an 18-state model
        write_tsv(ind / ("m_chr%d.txt" % (k + 1)), part, chrom="chr%d" % (k + 1))
    res, secs = _run(ind, tmp_path / "out", 3, timeout=300)
    assert res.returncode != 0
    assert "outside the 18-state model" in res.stdout + res.stderr
    assert secs < 120, secs
    assert not list((tmp_path / "out").glob("scores_*"))


def test_every_file_cut_by_a_border(tmp_path, golden_real, golden_pair):
    """Two equal files over three ranks: both are cut by a range border, the middle rank parses nothing and scores only rows
    that were handed over (driver._redistribute: a send / recv pair per border piece), single and paired."""
    for mode in ("single", "paired"):
        ind = tmp_path / ("in_" + mode)
        ind.mkdir()
        if mode == "single":
            for k in range(2):
                write_tsv(ind / ("m_chr%d.txt" % (k + 1)), golden_real["x"][k * 900:(k + 1) * 900], chrom="chr%d" % (k + 1))
            names = ["scores_t_s1_m_chr1.txt.gz", "scores_t_s1_m_chr2.txt.gz"]
        else:
            (ind / "A").mkdir(); (ind / "B").mkdir()
            for k in range(2):
                write_tsv(ind / "A" / ("m_chr%d.txt" % (k + 1)), golden_pair["xa"][k * 900:(k + 1) * 900], chrom="chr%d" % (k + 1))
                write_tsv(ind / "B" / ("m_chr%d.txt" % (k + 1)), golden_pair["xb"][k * 900:(k + 1) * 900], chrom="chr%d" % (k + 1))
            names = ["pairwiseDelta_t_s1_m_chr1.txt.gz", "pairwiseDelta_t_s1_m_chr2.txt.gz"]
        outs = {}
        for world in (1, 3):
            log = tmp_path / ("io_%s_%d.log" % (mode, world))
            res, _ = _run(ind, tmp_path / ("out_%s_%d" % (mode, world)), world, {"EPILOGOS_IO_LOG": str(log)})
            assert res.returncode == 0, res.stdout + res.stderr
            outs[world] = tmp_path / ("out_%s_%d" % (mode, world))
            lines = [l.split("\t") for l in log.read_text().splitlines()]
            assert len(lines) == (2 if mode == "single" else 4)       # every file parsed once, whatever the rank count
            if world == 3:
                assert len({l[0] for l in lines}) == 2                # ... by two of the three processes
        for name in names:
            assert _decompressed(outs[1] / name) == _decompressed(outs[3] / name), (mode, name)
        if mode == "paired":
            for stem in ("m_chr1", "m_chr2"):
                for kind, key in (("temp_nullDistances", "nullDistances"), ("temp_quiescence", "quiescenceArr")):
                    a = np.load(outs[1] / ("%s_t_s1_%s.npz" % (kind, stem)))[key]
                    b = np.load(outs[3] / ("%s_t_s1_%s.npz" % (kind, stem)))[key]
                    assert np.array_equal(a, b), (kind, stem)


def test_ranks_agree_on_the_plan_when_only_some_see_the_cache(tmp_path, golden_real, monkeypatch):
    """the "ranges" route (every row count known from the --cache-dir side-cars) and the "assigned" route run
    different collective sequences; a rank that sees the side-cars while another does not must not go its own way.  _plan takes
    the decision by an all-reduce: here through a stand-in communicator that reports what the OTHER rank saw."""
    from epilogos_amd import driver

    class Two:
        rank, world = 0, 2

        def __init__(self, other):
            self.other = other

        def max_ints(self, values):
            return [max(int(a), int(b)) for a, b in zip(values, self.other(values))]

    files = [tmp_path / "a.txt", tmp_path / "b.txt"]
    for f in files:
        f.write_text("chr1\t0\t200\t1\t2\n" * 10)
    monkeypatch.setattr(driver, "_cached_rows", lambda f: 10)
    mode, rows, _jobs, _owner = driver._plan(files, Two(lambda v: v), None)                    # both ranks see the same cache
    assert mode == "ranges" and rows == [10, 10]
    other_blind = lambda v: [1] + [0] * (len(v) - 1)                                           # the other rank sees no side-cars
    assert driver._plan(files, Two(other_blind), None)[0] == "assigned"
    other_rows = lambda v: [0, 10, 12, -10, -12]                                               # ... or different row counts
    assert driver._plan(files, Two(other_rows), None)[0] == "assigned"
    monkeypatch.setattr(driver, "_cached_rows", lambda f: 10 if f.name == "a.txt" else None)   # group 2 of a paired run uncached
    assert driver._plan(files[:1], Two(lambda v: v), None, files2=files[1:])[0] == "assigned"


def test_parse_assignment_is_balanced_on_hg19():
    """driver._assign_files shares the files out by bytes, longest first (LPT): on hg19's 24 chromosome sizes the fullest of 8
    ranks holds <= 1.15 x the mean (1.40 with the round-3 rule "the rank whose bin range holds most of the
    file"); every file has exactly one parser; one or two files per rank stay with the rank that owns their bins."""
    from epilogos_amd import driver
    import bench
    bp = bench.HG19_BP
    for world in (2, 3, 4, 8, 16):
        owner = driver._assign_files([None] * len(bp), world, sizes=bp)
        assert len(owner) == len(bp) and all(0 <= g < world for g in owner)
        load = np.bincount(owner, weights=bp, minlength=world)
        assert load.max() / load.mean() <= (1.15 if world <= 8 else 1.35), (world, load.max() / load.mean())
    assert driver._assign_files([None] * 2, 2, sizes=[100, 1000]) == [0, 1]
    assert driver._assign_files([None] * 3, 3, sizes=[500, 500, 500]) == [0, 1, 2]


def test_eight_ranks_twenty_four_files(tmp_path, golden_real):
    """The shape of the real job -- 24 chromosome files in hg19's proportions over EIGHT ranks (gloo, CPU stand-in backend):
    every file parsed once, by the rank the size-balanced assignment names, every range border inside a file handed over,
    outputs byte-identical to the one-rank run, exp_freq identical."""
    import bench
    from epilogos_amd import driver
    x = golden_real["x"]
    w = np.array(bench.HG19_BP, dtype=np.float64)
    edges = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * x.shape[0])]).astype(int)
    ind = tmp_path / "in"
    ind.mkdir()
    names = ["m_chr%02d.txt" % (k + 1) for k in range(24)]
    for k, name in enumerate(names):
        write_tsv(ind / name, x[edges[k]:edges[k + 1]], chrom="chr%d" % (k + 1))
    outs = {}
    for world in (1, 8):
        log = tmp_path / ("io%d.log" % world)
        res, _ = _run(ind, tmp_path / ("out%d" % world), world, {"EPILOGOS_IO_LOG": str(log)}, timeout=900)
        assert res.returncode == 0, res.stdout + res.stderr
        outs[world] = tmp_path / ("out%d" % world)
        lines = [l.split("\t") for l in log.read_text().splitlines()]
        assert sorted(l[2] for l in lines) == sorted(str(ind / n) for n in names)           # each file once in total
        if world == 8:
            by_pid = {}
            for l in lines:
                by_pid.setdefault(l[0], []).append(l[2])
            assert len(by_pid) == 8                                                        # every rank parsed something
            owner = driver._assign_files([ind / n for n in names], 8)
            groups = sorted(sorted(str(ind / names[k]) for k in range(24) if owner[k] == g) for g in range(8))
            assert sorted(sorted(v) for v in by_pid.values()) == groups                    # ... exactly what _assign_files gave it
    for name in names:
        stem = name[:-4]
        assert _decompressed(outs[1] / ("scores_t_s1_%s.txt.gz" % stem)) == _decompressed(outs[8] / ("scores_t_s1_%s.txt.gz" % stem)), stem
    assert np.array_equal(np.load(outs[1] / "exp_freq_t_s1.npy"), np.load(outs[8] / "exp_freq_t_s1.npy"))
    assert not list(outs[8].glob(".part_*"))

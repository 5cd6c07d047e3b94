"""bench.py's host-side pieces that run without a GPU: the CPU baseline leg (oracle/rowloop_baseline.py is test / baseline
infrastructure, never product code) and the shard arithmetic of the strong / weak split."""
import numpy as np

import bench


def test_cpu_baseline_fields_and_calibration():
    """One worker of the per-bin numpy loop must land in the range SURVEY.md 8d calibrated against the real reference
    (10.3 k bins/s/core on a 2.1 GHz Xeon; any host within 10x), and the line carries what the contract asks for."""
    cpu = bench.cpu_baseline(833, 18, target_seconds=1.5)
    assert cpu["unit"] == "Mbins/s" and cpu["kind"] == "port" and cpu["cores"] >= 1
    assert 1e3 < cpu["one_worker_bins_per_s"] < 1e5
    assert cpu["vectorised_numpy_one_core_bins_per_s"] > cpu["one_worker_bins_per_s"]      # the fairer second line is faster
    assert abs(cpu["value"] * 1e6 - cpu["per_core_bins_per_s"] * cpu["cores"]) <= 0.01 * cpu["value"] * 1e6 + 1
    assert "bin-scorings" in cpu["sample"]


def test_rowloop_baseline_equals_vectorised_oracle():
    from oracle import oracle_np as onp
    from oracle import rowloop_baseline as rb
    rng = np.random.default_rng(5)
    p = bench.FREQS / bench.FREQS.sum()
    x = rng.choice(18, size=(300, 61), p=p).astype(np.int64)
    q = onp.normalise(onp.expected_s1(x, 18))
    np.testing.assert_array_equal(rb.score_rows_s1(x, q, 18), onp.score_s1(x, q, 18).astype(np.float32))


def test_strong_split_is_the_reference_rule():
    from epilogos_amd.helpers import splitRows
    R = 15_000_000
    for world in (1, 2, 4, 8):
        mine = [(r * R // world, (r + 1) * R // world) for r in range(world)]       # bench.py's shard of rank r
        assert mine == [tuple(t) for t in splitRows(R, world)]


def test_gpus_flag_makes_bench_its_own_launcher(tmp_path):
    """`python bench.py --gpus N` from a plain start (no WORLD_SIZE): the process turns into the launcher -- a child
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`, started before torch is imported -- and
    returns the child's exit code (the reference's one command fans out by itself, run.py:190-279)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(bench.__file__).resolve().parent
    env = dict(os.environ, EPILOGOS_LAUNCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    words = res.stdout.split()
    assert words[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in words
    assert words[words.index("--nproc-per-node") + 1] == "4" and words[words.index("--master-addr") + 1] == "127.0.0.1"
    tail = words[words.index(str(root / "bench.py")):]
    assert tail[1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    # a child that fails makes the launcher fail: a world size that contradicts --gpus
    env.pop("EPILOGOS_LAUNCH_DRYRUN")
    bad = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "3"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "does not match" in bad.stderr


def test_chromosome_parts_cover_a_ranks_range_exactly():
    """bench.chromosome_parts (how bench.py feeds the paired session: one part per chromosome file in hg19's proportions, cut at
    the rank's bin range): the parts of all ranks tile the genome without gaps or overlaps, a part never crosses a file border,
    and (file, first row) -- what keys the null shuffle -- does not depend on the number of ranks."""
    from epilogos_amd.driver import shuffle_key
    R = 15_000_000
    whole = bench.chromosome_parts(R, 0, R)
    assert len(whole) == 24 and whole[0][2] == 0 and whole[-1][3] == R
    assert all(a[3] == b[2] for a, b in zip(whole, whole[1:])) and all(p[1] == 0 for p in whole)
    sizes = np.array([b - a for _f, _r, a, b in whole], dtype=np.float64)
    want = np.array(bench.HG19_BP, dtype=np.float64)
    assert np.abs(sizes / sizes.sum() - want / want.sum()).max() < 1e-6
    first = {f: a for f, _r, a, _b in whole}
    for world in (2, 3, 8):
        seen = []
        for rank in range(world):
            lo, hi = rank * R // world, (rank + 1) * R // world
            parts = bench.chromosome_parts(R, lo, hi)
            assert parts[0][2] == lo and parts[-1][3] == hi
            for f, r0, a, b in parts:
                assert a < b and r0 == a - first[f]                      # the row in the file of the part's first bin
                assert b <= first.get(f + 1, R)                          # never across a file border
            seen += parts
        assert all(a[3] == b[2] for a, b in zip(seen, seen[1:])) and seen[0][2] == 0 and seen[-1][3] == R
    # the key is (file, row): rows of one file ascend, files do not collide below 2^40 rows
    assert shuffle_key(3, 17) == (3 << 40) + 17 and shuffle_key(0, (1 << 40) - 1) < shuffle_key(1, 0)


def test_collective_selftest_two_gloo_ranks(tmp_path):
    """bench.rccl_selftest (every collective the multi-rank command line uses, checked against numbers) on two gloo ranks,
    host tensors: the leg that will explain the first real 8-GPU run must itself be right -- and must report a wrong
    result instead of raising."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    from tests.conftest import free_port
    root = Path(bench.__file__).resolve().parent
    script = tmp_path / "selftest.py"
    script.write_text(
        "import json, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "import bench\n"
        "dist.init_process_group(backend='gloo')\n"
        "rank, world = dist.get_rank(), dist.get_world_size()\n"
        "st = bench.rccl_selftest(torch, dist, torch.device('cpu'), rank, world, 61, 18)\n"
        "open(os.path.join(%r, 'st_%%d.json' %% rank), 'w').write(json.dumps(st))\n"
        "dist.destroy_process_group()\n" % (str(root), str(tmp_path)))
    env = dict(os.environ, PYTHONPATH=str(root))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr[-3000:]
    sts = [json.loads((tmp_path / ("st_%d.json" % r)).read_text()) for r in range(2)]
    for st in sts:
        assert st["ok"] and st["world"] == 2 and st["backend"] == "gloo"
        legs = [k for k, v in st.items() if isinstance(v, dict)]
        assert len(legs) == 6 and all(st[k]["ok"] for k in legs)


def test_line_keeper_prints_the_last_whole_line_however_its_parent_ends(tmp_path):
    """bench.LineKeeper (rank 0 of a multi-rank run): the child prints the latest line it was handed when the parent's end of the
    pipe closes -- after final(), after an abort, after a SIGTERM from the launcher -- and never a torn one."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    prog = """
import os, signal, sys, time
sys.path.insert(0, %r)
import bench
k = bench.LineKeeper()
k.keep('{"v": 1}')
k.keep('{"v": 2}')
how = sys.argv[1]
if how == "final":
    assert k.final('{"v": 3}')
elif how == "abort":
    os.abort()
elif how == "term":
    os.kill(os.getpid(), signal.SIGTERM)
    time.sleep(5)
elif how == "torn":
    k.p.stdin.write(b'{"v": 9')
    k.p.stdin.flush()
    os._exit(7)
""" % str(root)
    for how, want in (("final", '{"v": 3}'), ("abort", '{"v": 2}'), ("term", '{"v": 2}'), ("torn", '{"v": 2}')):
        res = subprocess.run([sys.executable, "-c", prog, how], capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
        assert (res.returncode == 0) == (how == "final"), (how, res.returncode, res.stderr[-500:])
        assert res.stdout.strip().splitlines() == [want], (how, res.stdout, res.stderr[-500:])

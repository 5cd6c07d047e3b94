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

"""SURVEY 8 row f4 -- STEP 4 of a paired run against vectors produced by the reference's roiAndVisualPairwise
functions (tests/golden/make_golden_pairwise.py).  CPU tests pin the oracle's per-bin reduction and the host logic
(through the oracle-backed fake backend); the GPU tests pin epg_pair_metrics and the whole STEP 4 through the C ABI."""
import gzip
from pathlib import Path

import numpy as np
import pytest
import scipy.stats as st

from epilogos_amd import _io
from epilogos_amd import roiAndVisualPairwise as rv
from oracle import oracle_np as onp
from tests.fake_backend import OracleBackend

GOLD = Path(__file__).parent / "golden"
S = 18


@pytest.fixture(scope="module")
def g():
    return dict(np.load(GOLD / "pairwise_step4.npz"))


def _state_info(tmp, names):
    p = tmp / "metadata.tsv"
    p.write_text("zero_index\tshort_name\n" + "".join("{}\t{}\n".format(i, n) for i, n in enumerate(names)))
    return p


def _stage(tmp, g, label, sidecar, backend):
    """The files STEP 3 leaves behind, one set per chromosome; returns the output directory."""
    d = tmp / ("out_" + label + ("_side" if sidecar else "_text"))
    d.mkdir()
    delta, start0 = g[label + "_delta"], int(g["start0"])
    for name, (lo, hi) in zip(g["split_names"], g["split_bounds"]):
        name = str(name)
        loc = np.array([[name, start0 + 200 * i, start0 + 200 * i + 200] for i in range(lo, hi)], dtype=object)
        _io.write_scores(d / "pairwiseDelta_t_matrix_{}.txt.gz".format(name), _io.Locations.from_object_array(loc), delta[lo:hi])
        np.savez_compressed(d / "temp_nullDistances_t_matrix_{}.npz".format(name), chrName=np.array([name]),
                            nullDistances=g["null_dist"][lo:hi])
        np.savez_compressed(d / "temp_quiescence_t_matrix_{}.npz".format(name), chrName=np.array([name]),
                            quiescenceArr=g["quiescent"][lo:hi])
        if sidecar:
            dist, md = backend.pair_metrics(delta[lo:hi], roundtrip=True)
            np.savez_compressed(d / "temp_pairMetrics_t_matrix_{}.npz".format(name), chrName=np.array([name]), distances=dist,
                                maxDiff=md, starts=loc[:, 1].astype(np.int64), ends=loc[:, 2].astype(np.int64))
    np.save(d / "exp_freq_t.npy", np.zeros(S, dtype=np.float32))
    return d


def _check_step4(tmp, g, label, pval, sidecar, backend):
    info = _state_info(tmp, g["state_names"])
    kind = "p" if pval else "z"
    for w in ((125, 10) if pval else (125, 10, 7)):
        sub = tmp / ("w%d" % w)
        sub.mkdir()
        d = _stage(sub, g, label, sidecar, backend)                   # main() consumes its inputs: stage per run
        rv.main("A", "B", info, d, "t", 1, pval, False, 3, 100000, d / "exp_freq_t.npy", w, False, backend=backend)
        assert not list(d.glob("temp_*.npz")) and not (d / "exp_freq_t.npy").exists()
        assert gzip.open(d / "pairwiseMetrics_t.txt.gz").read() == g["%s_metrics_%s" % (label, kind)].tobytes()
        assert (d / "regionsOfInterest_t.txt").read_bytes() == g["%s_roi_%s_w%d" % (label, kind, w)].tobytes()
        if pval:
            assert gzip.open(d / "significantLoci_t.txt.gz").read() == g[label + "_sig"].tobytes()


# ---------------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("label", ["real", "spiked"])
def test_oracle_pair_metrics_matches_reference(g, label):
    dist, md = onp.pair_metrics(g[label + "_delta"], roundtrip=True)
    assert dist.dtype == np.float32 and md.dtype == np.int32
    assert np.array_equal(dist, g[label + "_dist"])
    assert np.array_equal(md, g[label + "_maxdiff"])
    # values that already went through the text are a fixed point of the round trip
    parsed = onp.text_roundtrip_f5(g[label + "_delta"])
    d2, m2 = onp.pair_metrics(parsed, roundtrip=False)
    assert np.array_equal(d2, dist) and np.array_equal(m2, md)


def test_text_roundtrip_is_what_the_parser_reads(g):
    v = g["spiked_delta"][:64]
    txt = ["{:.5f}".format(x) for x in v.ravel()]
    assert np.array_equal(onp.text_roundtrip_f5(v).ravel(), np.array([float(t) for t in txt]).astype(np.float32))


def test_benjamini_hochberg(g):
    for label in ("real", "spiked"):
        mh = rv.benjaminiHochberg(g[label + "_pvals"])
        np.testing.assert_allclose(mh, g[label + "_mh"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(mh, st.false_discovery_control(g[label + "_pvals"], method="bh"), rtol=1e-12, atol=0)
    assert rv.benjaminiHochberg(np.array([])).shape == (0,)
    np.testing.assert_allclose(rv.benjaminiHochberg(np.array([0.01, 0.04, 0.03, 0.9])), [0.04, 0.16 / 3, 0.16 / 3, 0.9])


def test_pvalues_and_fit(g, tmp_path):
    for label in ("real", "spiked"):
        beta, loc, scale = g[label + "_params"]
        np.testing.assert_array_equal(rv.calculatePVals(g[label + "_dist"], beta, loc, scale), g[label + "_pvals"])
    d = _stage(tmp_path, g, "real", False, OracleBackend())
    params, null, nonq = rv.fitDistances(d, 1, 3, 100000)
    np.testing.assert_allclose(params, g["real_params"], rtol=1e-9)
    assert np.array_equal(null, g["null_dist"]) and np.array_equal(nonq, np.where(~g["quiescent"])[0])


@pytest.mark.parametrize("sidecar", [True, False])
@pytest.mark.parametrize("pval", [False, True])
@pytest.mark.parametrize("label", ["real", "spiked"])
def test_step4_host_logic(g, tmp_path, label, pval, sidecar):
    _check_step4(tmp_path, g, label, pval, sidecar, OracleBackend())


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("label", ["real", "spiked"])
def test_hip_pair_metrics_bit_exact(g, label):
    import torch
    from epilogos_amd import engine
    engine.require_gpu()
    delta = torch.from_numpy(g[label + "_delta"]).cuda()
    dist, md = engine.pair_metrics(delta, roundtrip=True)
    assert np.array_equal(dist.cpu().numpy(), g[label + "_dist"])
    assert np.array_equal(md.cpu().numpy(), g[label + "_maxdiff"])
    parsed = torch.from_numpy(onp.text_roundtrip_f5(g[label + "_delta"])).cuda()
    d2, m2 = engine.pair_metrics(parsed, roundtrip=False)
    assert torch.equal(d2, dist) and torch.equal(m2, md)


@pytest.mark.gpu
def test_hip_pair_metrics_random_against_oracle():
    import torch
    from epilogos_amd import engine
    rng = np.random.default_rng(3)
    for R, S_ in ((1, 18), (1000, 15), (4097, 25), (70001, 18)):
        d = (rng.normal(0, 1, (R, S_)) * rng.choice([1e-6, 1e-3, 1, 50], (R, 1))).astype(np.float32)
        d[rng.random((R, S_)) < 0.1] = 0
        d[::7] = np.round(d[::7], 5) + np.float32(5e-6)                  # decimal ties and near-ties
        if R > 10:
            d[3] = 0                                                      # all zero: sign 0, last state wins
            d[4] = -d[4, ::-1].copy()
        dist, md = engine.pair_metrics(torch.from_numpy(d).cuda(), roundtrip=True)
        rd, rm = onp.pair_metrics(d, roundtrip=True)
        assert np.array_equal(dist.cpu().numpy(), rd) and np.array_equal(md.cpu().numpy(), rm)
    with pytest.raises(Exception):
        engine._abi.call("epg_pair_metrics", None, 5, 18, 1, None, None, None)


@pytest.mark.gpu
@pytest.mark.parametrize("sidecar", [True, False])
@pytest.mark.parametrize("pval", [False, True])
def test_step4_through_the_abi(g, tmp_path, pval, sidecar):
    from epilogos_amd import backend
    _check_step4(tmp_path, g, "spiked", pval, sidecar, backend.HipBackend())


def test_location_order_shortcut_is_exactly_the_stable_sort():
    """mainFromArrays skips the three-key lexsort of reference :332 when it would be the identity; the predicate must say so
    exactly then (ties on all three keys included: the sort is stable)."""
    from epilogos_amd.roiAndVisualPairwise import _is_location_sorted
    rng = np.random.default_rng(3)
    for trial in range(200):
        n = int(rng.integers(0, 40))
        c = np.sort(rng.integers(1, 4, n)).astype(np.int64)
        s = rng.integers(0, 5, n).astype(np.int64) * 200
        e = s + rng.integers(0, 3, n) * 200
        if trial % 2 == 0 and n:                                    # make it sorted
            o = np.lexsort((e, s, c))
            c, s, e = c[o], s[o], e[o]
        identity = np.array_equal(np.lexsort((e, s, c)), np.arange(n))
        assert _is_location_sorted(c, s, e) == identity, (c, s, e)

"""GPU parity for the S3 kernels and the paired-mode null shuffle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle_np as onp
from tests.conftest import synth_states

pytestmark = pytest.mark.gpu
S = 18


@pytest.fixture(scope="module")
def eng():
    from epilogos_amd import engine
    engine.require_gpu()
    return engine


def _np(t):
    return t.cpu().numpy()


def _s3_counts(eng, x):
    N = x.shape[1]
    X = eng.states_to_device(x)
    c = eng.hist_s3(X, N, S)
    return c, _np(c).reshape(N, N, S, S)


def test_s3_counts_golden(eng, golden_s3, golden_real, golden_edge):
    for x, want in ((golden_s3["x"], golden_s3["s3_counts"]), (golden_real["x"], golden_real["s3_counts"]),
                    (golden_edge["n2_x"], golden_edge["n2_s3_counts"])):
        c, got = _s3_counts(eng, x)
        assert got.dtype == np.int32 and np.array_equal(got, want)
        N = x.shape[1]
        assert np.array_equal(_np(eng.normalise(c)).reshape(N, N, S, S), onp.normalise(want))


def test_s3_counts_exp_golden(eng, golden_s3):
    c, got = _s3_counts(eng, golden_s3["x"])
    assert np.array_equal(_np(eng.normalise(c)).reshape(got.shape), golden_s3["s3_exp"])   # bit-exact float32 exp_freq


@pytest.mark.parametrize("N,R", [(70, 500), (65, 130), (3, 70000), (129, 64)])
def test_s3_counts_random(eng, N, R):
    x = synth_states(R, N, seed=N)
    x[0, 0] = -1                                   # an invalid state is skipped in every pair it takes part in
    c, got = _s3_counts(eng, x)
    xo = x.copy()
    want = onp.expected_s3(xo, S)
    assert np.array_equal(got, want)
    X = eng.states_to_device(x)
    eng.hist_s3(X, N, S, counts=c)                 # accumulates
    assert np.array_equal(_np(c).reshape(got.shape), 2 * want)


def _s3_score_check(eng, x, q, gold32=None):
    N = x.shape[1]
    X = eng.states_to_device(x)
    qd = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32).reshape(-1)).cuda()
    o32, o64 = eng.score_s3(X, N, S, qd, want32=True, want64=True)
    ref = onp.score_s3_f64(x, q, S)
    # the contract's tolerance (north_star: 1e-6 relative): float32 table with every operation correctly rounded, exact
    # fixed-point sums -- observed <= 1.4e-7 over all shapes of this suite (profiles/r06a_s3_score_precision.txt; 1.8e-7 with the
    # device's float32 log2f of rounds 1-5, whose tests asserted 2e-6)
    np.testing.assert_allclose(_np(o64), ref, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(_np(o32), ref.astype(np.float32), rtol=1e-6, atol=1e-9)
    if gold32 is not None:   # the reference's float32 sequential accumulation (SURVEY 8c tolerance)
        np.testing.assert_allclose(_np(o32), gold32, rtol=1e-4, atol=5e-6)
    o32b, _ = eng.score_s3(X, N, S, qd, want32=True, want64=False)
    assert torch.equal(o32b, o32)             # fixed-point cells: the order of the blocks' atomics does not matter


def test_s3_score_golden(eng, golden_s3, golden_real, golden_edge):
    _s3_score_check(eng, golden_s3["x"], golden_s3["s3_exp"], golden_s3["s3_f32"])
    _s3_score_check(eng, golden_real["x"][:600], golden_real["s3_exp"], golden_real["s3_f32"][:600])
    _s3_score_check(eng, golden_edge["n2_x"], golden_edge["n2_s3_exp"], golden_edge["n2_s3_f32"])


@pytest.mark.parametrize("N", [33, 65])
def test_s3_score_tile_not_multiple_of_four(eng, N):
    """N*S % 4 != 0: the LDS tile is loaded with scalar loads, and R is not a multiple of the MFMA K step."""
    x = synth_states(211, N, seed=N)
    c = onp.expected_s3(x, S)
    _, got = _s3_counts(eng, x)
    assert np.array_equal(got, c)
    _s3_score_check(eng, x, onp.normalise(c))


@pytest.mark.parametrize("S_,N,R", [(5, 40, 300), (13, 23, 1000), (15, 64, 700), (25, 30, 257), (30, 21, 420), (31, 9, 300),
                                     (15, 200, 1500), (20, 70, 1500), (19, 33, 2900), (21, 40, 300)])
def test_s3_other_state_counts(eng, S_, N, R):
    """State models other than 18: S <= 30 takes the one-hot fp4 contraction, S = 31 the LDS-atomic kernel; k_s3_score (S > 20)
    is instantiated for 6, 10 and 16 staging elements per thread, k_s3_score_bl (S <= 20) for 16, 32, 43 and 53 table loads per
    chunk (S = 5 / 13 / 15, 18 / 19, 20), with one to seven chunks of 32 biosamples and one to three bin slices of 1440."""
    x = synth_states(R, N, S=S_, seed=S_, uniform=True)
    keep = x[R // 2, N // 2]
    x[R // 2, N // 2] = -1                         # not a state: skipped in every pair it takes part in
    want = onp.expected_s3(x, S_)
    c = eng.hist_s3(eng.states_to_device(x), N, S_)
    assert np.array_equal(_np(c).reshape(want.shape), want)
    x[R // 2, N // 2] = keep                       # scores are defined for valid states only
    X = eng.states_to_device(x)
    q = onp.normalise(want)
    qd = torch.from_numpy(q.reshape(-1)).cuda()
    o32, o64 = eng.score_s3(X, N, S_, qd, want32=True, want64=True)
    ref = onp.score_s3_f64(x, q, S_)
    np.testing.assert_allclose(_np(o64), ref, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(_np(o32), ref.astype(np.float32), rtol=1e-6, atol=1e-9)


def test_s3_score_random(eng):
    x = synth_states(9000, 70, seed=3)             # more than one slice of 8192 bins
    q = onp.normalise(onp.expected_s3(x, S))
    _s3_score_check(eng, x[:300], q)
    X = eng.states_to_device(x)
    qd = torch.from_numpy(q.reshape(-1)).cuda()
    o32, o64 = eng.score_s3(X, 70, S, qd, want32=True, want64=True)
    ref_tail = onp.score_s3_f64(x[8100:8300], q, S)
    np.testing.assert_allclose(_np(o64)[8100:8300], ref_tail, rtol=1e-6, atol=1e-9)


# ------------------------------------------------------------------------------------------------ null shuffle
def test_null_hist_properties(eng):
    R, NA, NB = 4000, 37, 29
    xa = synth_states(R, NA, seed=1)
    xb = synth_states(R, NB, seed=2)
    XA, XB = eng.states_to_device(xa), eng.states_to_device(xb)
    tot = onp.bin_hist(np.concatenate([xa, xb], axis=1), S)
    HA, HB = eng.null_hist(XA, NA, XB, NB, S, NA, NB, seed=123)
    ha, hb = eng.hist_to_numpy(HA).astype(np.int64), eng.hist_to_numpy(HB).astype(np.int64)
    assert (ha.sum(axis=1) == NA).all() and (hb.sum(axis=1) == NB).all()
    assert np.array_equal(ha + hb, tot)            # a permutation: every column lands in exactly one group
    # reproducible, seed-dependent, independent of which slice of rows a call covers (row0 offsets)
    HA2, HB2 = eng.null_hist(XA, NA, XB, NB, S, NA, NB, seed=123)
    assert torch.equal(HA, HA2) and torch.equal(HB, HB2)
    HA3, _ = eng.null_hist(XA, NA, XB, NB, S, NA, NB, seed=124)
    assert not torch.equal(HA, HA3)
    XAs, XBs = eng.states_to_device(xa[1000:1500]), eng.states_to_device(xb[1000:1500])
    HAs, HBs = eng.null_hist(XAs, NA, XBs, NB, S, NA, NB, seed=123, row0=1000)
    assert torch.equal(HAs, HA[1000:1500]) and torch.equal(HBs, HB[1000:1500])
    # group-size option: two groups of g columns drawn without replacement
    g = 20
    HAg, HBg = eng.null_hist(XA, NA, XB, NB, S, g, g, seed=5)
    hag, hbg = eng.hist_to_numpy(HAg).astype(np.int64), eng.hist_to_numpy(HBg).astype(np.int64)
    assert (hag.sum(axis=1) == g).all() and (hbg.sum(axis=1) == g).all() and (hag + hbg <= tot).all()


@pytest.mark.parametrize("packed", [False, True])
def test_null_hist_wide_rows(eng, packed):
    """The 379 + 342 split of the headline matrix: several 64-column chunks per group, a last partial chunk, a Philox
    block cut by the end of a group (379 = 4 * 94 + 3), rows that do not fill the last block, bytes that are not states,
    and rows at pitch N (pieces that cross the end of a row go through the byte path)."""
    R, NA, NB = 1000, 379, 342
    xa = synth_states(R, NA, seed=11)
    xb = synth_states(R, NB, seed=12)
    xa[5, 3] = -1
    xb[7, 341] = 25
    if packed:
        XA = torch.from_numpy(xa.copy()).cuda()
        XB = torch.from_numpy(xb.copy()).cuda()
    else:
        XA, XB = eng.states_to_device(xa), eng.states_to_device(xb)
    tot = onp.bin_hist(np.concatenate([xa, xb], axis=1), S).astype(np.int64)
    HA, HB = eng.null_hist(XA, NA, XB, NB, S, NA, NB, seed=7)
    ha, hb = eng.hist_to_numpy(HA).astype(np.int64), eng.hist_to_numpy(HB).astype(np.int64)
    assert np.array_equal(ha + hb, tot)
    ok = np.ones(R, dtype=bool); ok[[5, 7]] = False
    assert (ha[ok].sum(axis=1) == NA).all() and (hb[ok].sum(axis=1) == NB).all()
    assert ha[5].sum() + hb[5].sum() == NA + NB - 1 and ha[7].sum() + hb[7].sum() == NA + NB - 1
    # the same rows through the padded layout give the same draws
    HA2, HB2 = eng.null_hist(eng.states_to_device(xa), NA, eng.states_to_device(xb), NB, S, NA, NB, seed=7)
    assert torch.equal(HA, HA2) and torch.equal(HB, HB2)
    # mean of group A's count of the dominant state = NA / (NA + NB) of the row's total
    frac = ha[:, 17].sum() / tot[:, 17].sum()
    assert abs(frac - NA / (NA + NB)) < 0.005
    HAg, HBg = eng.null_hist(XA, NA, XB, NB, S, 100, 100, seed=7)
    hag, hbg = eng.hist_to_numpy(HAg).astype(np.int64), eng.hist_to_numpy(HBg).astype(np.int64)
    assert (hag[ok].sum(axis=1) == 100).all() and (hbg[ok].sum(axis=1) == 100).all() and (hag + hbg <= tot).all()


def test_null_hist_is_a_uniform_shuffle(eng):
    """Same row repeated: the count of state s in group A is hypergeometric(M, K_s, NA) -- check mean and variance,
    and compare with numpy's argsort-of-uniforms shuffle (the reference's method, helpers.py:183-184)."""
    R, NA, NB = 60000, 12, 9
    M = NA + NB
    base = np.array([0] * 8 + [5] * 6 + [17] * 7, dtype=np.int8)
    xa = np.tile(base[:NA], (R, 1))
    xb = np.tile(base[NA:], (R, 1))
    HA, _ = eng.null_hist(eng.states_to_device(xa), NA, eng.states_to_device(xb), NB, S, NA, NB, seed=99)
    ha = eng.hist_to_numpy(HA).astype(np.float64)
    rng = np.random.default_rng(0)
    comb = np.tile(base, (R, 1))
    sh = onp.shuffle_rows(comb, rng.random(comb.shape))[:, :NA]
    for s, K in ((0, 8), (5, 6), (17, 7)):
        mean = NA * K / M
        var = NA * (K / M) * (1 - K / M) * (M - NA) / (M - 1)
        assert abs(ha[:, s].mean() - mean) < 5 * np.sqrt(var / R)
        assert abs(ha[:, s].var() - var) < 0.05 * var
        ref = (sh == s).sum(axis=1)
        # same distribution as the reference's shuffle: compare the empirical pmf
        pm_gpu = np.bincount(ha[:, s].astype(int), minlength=NA + 1) / R
        pm_ref = np.bincount(ref, minlength=NA + 1) / R
        assert np.abs(pm_gpu - pm_ref).max() < 0.01


def test_paired_pipeline_hip(tmp_path, golden_pair):
    import gzip
    from epilogos_amd import expected, expectedCombination, scores
    from tests.test_host_logic import write_tsv
    g = golden_pair
    a, b, out = tmp_path / "A", tmp_path / "B", tmp_path / "out"
    for d in (a, b, out):
        d.mkdir()
    write_tsv(a / "matrix_chr1.txt", g["xa"]); write_tsv(b / "matrix_chr1.txt", g["xb"])
    for sal in (1, 2):
        tag = "A_B_s%d" % sal
        expected.main(a / "matrix_chr1.txt", b / "matrix_chr1.txt", S, sal, out, tag, 1, False)
        expectedCombination.main(out, out / ("exp_freq_%s.npy" % tag), tag, False)
        assert np.array_equal(np.load(out / ("exp_freq_%s.npy" % tag)), g["s%d_exp" % sal])
        scores.NULL_SEED = 11
        scores.main(a / "matrix_chr1.txt", b / "matrix_chr1.txt", S, sal, out, out / ("exp_freq_%s.npy" % tag), tag, 1,
                    int(g["qstate"]), -1, False)
        with gzip.open(out / ("pairwiseDelta_%s_matrix_chr1.txt.gz" % tag), "rt") as fh:
            delta = np.array([[float(v) for v in l.rstrip("\n").split("\t")[3:]] for l in fh], dtype=np.float32)
        np.testing.assert_allclose(delta, g["s%d_delta" % sal], atol=1.01e-5)
        assert np.array_equal(np.load(out / ("temp_quiescence_%s_matrix_chr1.npz" % tag))["quiescenceArr"], g["s1_quiescent"])
        nd = np.load(out / ("temp_nullDistances_%s_matrix_chr1.npz" % tag))["nullDistances"]
        gd = g["s%d_null_dist" % sal]
        assert nd.shape == gd.shape and nd.dtype == np.float32
        # unseeded in the reference: same scale and sign balance, not the same draws
        assert 0.5 < (np.abs(nd).mean() + 1e-12) / (np.abs(gd).mean() + 1e-12) < 2.0


def test_null_hist_from_binhist_properties(eng):
    """The histogram-based sampler (epg_null_hist_from_binhist): a permutation's bookkeeping, reproducibility, independence
    of the row range a call covers, the group-size option, and columns that hold no state."""
    R, NA, NB = 4000, 379, 342
    xa = synth_states(R, NA, seed=21)
    xb = synth_states(R, NB, seed=22)
    xa[5, 3] = -1                                   # a column without a state takes part in the shuffle, is not reported
    xb[7, 341] = 25
    XA, XB = eng.states_to_device(xa), eng.states_to_device(xb)
    HA, _ = eng.bin_hist(XA, NA, S, want_counts=False)
    HB, _ = eng.bin_hist(XB, NB, S, want_counts=False)
    tot = (eng.hist_to_numpy(HA).astype(np.int64) + eng.hist_to_numpy(HB).astype(np.int64))
    OA, OB = eng.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=123)
    oa, ob = eng.hist_to_numpy(OA).astype(np.int64), eng.hist_to_numpy(OB).astype(np.int64)
    assert np.array_equal(oa + ob, tot)            # every column with a state lands in exactly one group
    ok = np.ones(R, dtype=bool); ok[[5, 7]] = False
    assert (oa[ok].sum(axis=1) == NA).all() and (ob[ok].sum(axis=1) == NB).all()
    assert oa[5].sum() + ob[5].sum() == NA + NB - 1 and oa[7].sum() + ob[7].sum() == NA + NB - 1
    OA2, OB2 = eng.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=123)
    assert torch.equal(OA, OA2) and torch.equal(OB, OB2)
    OA3, _ = eng.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=124)
    assert not torch.equal(OA, OA3)
    OAs, OBs = eng.null_hist_from_binhist(HA[1024:1600].contiguous(), HB[1024:1600].contiguous(), NA + NB, S, NA, NB, seed=123, row0=1024)
    assert torch.equal(OAs, OA[1024:1600]) and torch.equal(OBs, OB[1024:1600])
    frac = oa[:, 17].sum() / tot[:, 17].sum()      # the dominant state is never drawn, only completed: still NA / (NA + NB)
    assert abs(frac - NA / (NA + NB)) < 0.002
    OAg, OBg = eng.null_hist_from_binhist(HA, HB, NA + NB, S, 100, 100, seed=7)
    oag, obg = eng.hist_to_numpy(OAg).astype(np.int64), eng.hist_to_numpy(OBg).astype(np.int64)
    assert (oag[ok].sum(axis=1) == 100).all() and (obg[ok].sum(axis=1) == 100).all() and (oag + obg <= tot).all()
    # odd state counts / tiny shapes
    for S_, R_, na, nb in ((15, 70, 9, 4), (5, 1, 3, 3), (31, 129, 40, 41)):
        ya, yb = synth_states(R_, na, S=S_, seed=S_, uniform=True), synth_states(R_, nb, S=S_, seed=S_ + 1, uniform=True)
        Ha, _ = eng.bin_hist(eng.states_to_device(ya), na, S_, want_counts=False)
        Hb, _ = eng.bin_hist(eng.states_to_device(yb), nb, S_, want_counts=False)
        Oa, Ob = eng.null_hist_from_binhist(Ha, Hb, na + nb, S_, na, nb, seed=3)
        t = eng.hist_to_numpy(Ha).astype(np.int64) + eng.hist_to_numpy(Hb).astype(np.int64)
        assert np.array_equal(eng.hist_to_numpy(Oa).astype(np.int64) + eng.hist_to_numpy(Ob).astype(np.int64), t)
        assert (eng.hist_to_numpy(Oa).astype(np.int64).sum(axis=1) == na).all()


def test_null_hist_from_binhist_is_a_uniform_shuffle(eng):
    """Same row repeated: the histogram-based sampler gives the counts of a uniform shuffle -- hypergeometric mean and variance
    per state (the dominant state, which is never drawn, included), the joint law of two states, and the same empirical pmf as
    numpy's argsort-of-uniforms shuffle (the reference's method, helpers.py:183-184) and as the column-based kernel."""
    R, NA, NB = 60000, 12, 9
    M = NA + NB
    base = np.array([0] * 8 + [5] * 6 + [17] * 7, dtype=np.int8)
    xa = np.tile(base[:NA], (R, 1))
    xb = np.tile(base[NA:], (R, 1))
    XA, XB = eng.states_to_device(xa), eng.states_to_device(xb)
    HA, _ = eng.bin_hist(XA, NA, S, want_counts=False)
    HB, _ = eng.bin_hist(XB, NB, S, want_counts=False)
    OA, _ = eng.null_hist_from_binhist(HA, HB, M, S, NA, NB, seed=99)
    oa = eng.hist_to_numpy(OA).astype(np.float64)
    CA, _ = eng.null_hist(XA, NA, XB, NB, S, NA, NB, seed=5)
    ca = eng.hist_to_numpy(CA).astype(np.int64)
    rng = np.random.default_rng(0)
    comb = np.tile(base, (R, 1))
    sh = onp.shuffle_rows(comb, rng.random(comb.shape))[:, :NA]
    for s, K in ((0, 8), (5, 6), (17, 7)):
        mean = NA * K / M
        var = NA * (K / M) * (1 - K / M) * (M - NA) / (M - 1)
        assert abs(oa[:, s].mean() - mean) < 5 * np.sqrt(var / R)
        assert abs(oa[:, s].var() - var) < 0.05 * var
        pm_new = np.bincount(oa[:, s].astype(int), minlength=NA + 1) / R
        pm_ref = np.bincount((sh == s).sum(axis=1), minlength=NA + 1) / R
        pm_col = np.bincount(ca[:, s], minlength=NA + 1) / R
        assert np.abs(pm_new - pm_ref).max() < 0.01 and np.abs(pm_new - pm_col).max() < 0.01
    # joint law: cov(X_0, X_5) = -n (K0/M)(K5/M)(M - n)/(M - 1)
    cov = np.cov(oa[:, 0], oa[:, 5])[0, 1]
    want = -NA * (8 / M) * (6 / M) * (M - NA) / (M - 1)
    assert abs(cov - want) < 0.05 * abs(want)


def test_null_hist_from_binhist_large_categories_follow_the_hypergeometric_law(eng):
    """The law of the histogram-based sampler on a wide row with large categories: same row repeated with 70, 45, 30, 9 and 3
    columns per state plus 2 stateless ones -- means, variances, covariances within a group and across the two groups, and the
    empirical pmfs against numpy's argsort shuffle (the reference's method, helpers.py:183-184), for the default group sizes and
    for -g 40 (a third outcome: a column joins neither group).  (Written for a round-3 experiment that drew categories above 16
    columns by hypergeometric inversion from the mode instead of column by column: exact and no faster -- a float64 division per
    step of the walk costs what ten column draws do, and a loop per category makes a wave wait for its largest count in every
    category -- so the column-by-column sampler stayed)"""
    R, NA, NB = 40000, 90, 69
    M = NA + NB
    base = np.array([17] * 70 + [5] * 45 + [4] * 30 + [0] * 9 + [11] * 3 + [-1] * 2, dtype=np.int8)
    assert base.size == M
    rng = np.random.default_rng(1)
    cols = rng.permutation(M)                                       # which group a column starts in does not matter
    xa = np.tile(base[cols[:NA]], (R, 1))
    xb = np.tile(base[cols[NA:]], (R, 1))
    XA, XB = eng.states_to_device(xa), eng.states_to_device(xb)
    HA, _ = eng.bin_hist(XA, NA, S, want_counts=False)
    HB, _ = eng.bin_hist(XB, NB, S, want_counts=False)
    comb = np.tile(base, (R, 1))
    for ga, gb in ((NA, NB), (40, 40)):
        OA, OB = eng.null_hist_from_binhist(HA, HB, M, S, ga, gb, seed=2024)
        oa, ob = eng.hist_to_numpy(OA).astype(np.float64), eng.hist_to_numpy(OB).astype(np.float64)
        sh = onp.shuffle_rows(comb, np.random.default_rng(ga).random(comb.shape))
        ra, rb = sh[:, :ga], sh[:, ga:ga + gb]
        for s, K in ((17, 70), (5, 45), (4, 30), (0, 9), (11, 3)):
            for o, r, n in ((oa, ra, ga), (ob, rb, gb)):
                mean = n * K / M
                var = n * (K / M) * (1 - K / M) * (M - n) / (M - 1)
                assert abs(o[:, s].mean() - mean) < 5 * np.sqrt(var / R), (s, ga)
                assert abs(o[:, s].var() - var) < 0.06 * var, (s, ga)
                pm_new = np.bincount(o[:, s].astype(int), minlength=n + 1) / R
                pm_ref = np.bincount((r == s).sum(axis=1), minlength=n + 1) / R
                assert np.abs(pm_new - pm_ref).max() < 0.012, (s, ga)
        # joint laws: two large categories within A; one category across A and B
        want = -ga * (70 / M) * (45 / M) * (M - ga) / (M - 1)
        assert abs(np.cov(oa[:, 17], oa[:, 5])[0, 1] - want) < 0.06 * abs(want)
        want_ab = -ga * gb * (70 / M) * (1 - 70 / M) / (M - 1)
        assert abs(np.cov(oa[:, 17], ob[:, 17])[0, 1] - want_ab) < 0.06 * abs(want_ab)
        assert (oa.sum(axis=1) + (oa[:, :0].sum(axis=1)) <= ga).all() and (oa + ob <= np.bincount(base[base >= 0], minlength=S)).all()


def test_null_hist_bit_string_kernel_equals_the_column_by_column_kernel(eng, monkeypatch):
    """Round 3 took the state bookkeeping out of the histogram-based sampler's draw loop (one outcome bit per position, range
    popcounts afterwards).  The draws themselves -- Philox counters, one byte per position, the tie rule -- are those of the
    round-2 kernel (epg_test_force(0, 1)), so the same seed must give the SAME null groups: default group sizes, -g, columns
    without a state, wide rows, narrow rows, rows of one state, a 31-state model."""
    rng = np.random.default_rng(8)
    for S_, R_, na, nb, ga, gb in ((S, 5000, 379, 342, 379, 342), (S, 3000, 379, 342, 100, 100), (15, 777, 65, 62, 65, 62),
                                   (31, 300, 40, 41, 40, 41), (5, 130, 3, 3, 3, 3), (S, 600, 700, 690, 700, 690), (S, 64, 12, 9, 5, 7)):
        ya = synth_states(R_, na, S=S_, seed=int(rng.integers(1 << 30)))
        yb = synth_states(R_, nb, S=S_, seed=int(rng.integers(1 << 30)))
        ya[R_ // 2, 0] = -1                                  # a column without a state
        ya[R_ // 3, :] = 1                                   # a bin in which every column of A holds one state ...
        yb[R_ // 3, :] = 1                                   # ... and of B too: nothing to draw
        Ha, _ = eng.bin_hist(eng.states_to_device(ya), na, S_, want_counts=False)
        Hb, _ = eng.bin_hist(eng.states_to_device(yb), nb, S_, want_counts=False)
        Oa, Ob = eng.null_hist_from_binhist(Ha, Hb, na + nb, S_, ga, gb, seed=77, row0=123456789012)
        eng._abi.call("epg_test_force", 0, 1)                # the column-by-column kernel (what rows beyond 3072 columns take)
        try:
            Qa, Qb = eng.null_hist_from_binhist(Ha, Hb, na + nb, S_, ga, gb, seed=77, row0=123456789012)
        finally:
            eng._abi.call("epg_test_force", 0, 0)
        assert torch.equal(Oa, Qa) and torch.equal(Ob, Qb), (S_, R_, na, nb, ga, gb)
        assert (eng.hist_to_numpy(Oa).astype(np.int64).sum(axis=1)[np.arange(R_) != R_ // 2] == ga).all()


@pytest.mark.parametrize("na,nb", [(16, 16), (17, 16), (1500, 1572), (1600, 1600), (3000, 3001)])
def test_null_hist_from_binhist_row_widths_around_the_kernel_switch(eng, monkeypatch, na, nb):
    """Row widths at the word boundaries of the bit string (32, 33 columns), at the widest row the bit-string kernel takes
    (3072 columns: 24 KB of outcome bits per wave) and beyond it, where the call goes to the column-by-column kernel by itself:
    a permutation's bookkeeping in every case, and the same groups from both kernels wherever both apply."""
    R_ = 200
    ya = synth_states(R_, na, seed=na)
    yb = synth_states(R_, nb, seed=nb + 7)
    Ha, _ = eng.bin_hist(eng.states_to_device(ya), na, S, want_counts=False)
    Hb, _ = eng.bin_hist(eng.states_to_device(yb), nb, S, want_counts=False)
    tot = eng.hist_to_numpy(Ha).astype(np.int64) + eng.hist_to_numpy(Hb).astype(np.int64)
    Oa, Ob = eng.null_hist_from_binhist(Ha, Hb, na + nb, S, na, nb, seed=5)
    oa, ob = eng.hist_to_numpy(Oa).astype(np.int64), eng.hist_to_numpy(Ob).astype(np.int64)
    assert np.array_equal(oa + ob, tot) and (oa.sum(axis=1) == na).all() and (ob.sum(axis=1) == nb).all()
    eng._abi.call("epg_test_force", 0, 1)
    try:
        Qa, Qb = eng.null_hist_from_binhist(Ha, Hb, na + nb, S, na, nb, seed=5)
    finally:
        eng._abi.call("epg_test_force", 0, 0)
    assert torch.equal(Oa, Qa) and torch.equal(Ob, Qb)

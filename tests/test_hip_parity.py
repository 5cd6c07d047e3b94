"""GPU parity: the HIP kernels (through the C ABI) against the oracle and the reference's golden vectors.
Integers bit-exact; float64 KL scores within 1e-6 relative (BASELINE.json north_star) -- asserted much tighter."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle_np as onp
from tests.conftest import synth_states

pytestmark = pytest.mark.gpu

S = 18
RTOL = 1e-6          # the bar north_star states for float64 KL scores
RTOL_TIGHT = 1e-11   # what the kernels actually achieve (device log2 vs numpy log2)
ATOL = 1e-12


@pytest.fixture(scope="module")
def eng():
    from epilogos_amd import engine
    engine.require_gpu()
    return engine


def _np(t):
    return t.cpu().numpy()


def _unaligned_states(x, offset=1):
    """Device view with ldx == N and a base pointer that is not 16-byte aligned."""
    R, N = x.shape
    flat = torch.full((R * N + offset + 64,), -1, dtype=torch.int8, device="cuda")
    flat[offset:offset + R * N] = torch.from_numpy(np.ascontiguousarray(x).astype(np.int8)).reshape(-1).cuda()
    return flat[offset:offset + R * N].view(R, N)


# ---------------------------------------------------------------------------------------------- K1 histogram
@pytest.mark.parametrize("N", [1, 2, 10, 15, 16, 17, 31, 33, 127, 128, 129, 379, 833, 896, 897, 1100])
@pytest.mark.parametrize("R", [1, 17, 1000])
def test_bin_hist_shapes(eng, N, R):
    x = synth_states(R, N, seed=N * 7 + R)
    X = eng.states_to_device(x)
    H, counts = eng.bin_hist(X, N, S)
    h_ref = onp.bin_hist(x, S)
    assert np.array_equal(eng.hist_to_numpy(H).astype(np.int64), h_ref)
    assert np.array_equal(_np(counts), h_ref.sum(axis=0))
    # counts accumulate (+=), like the per-file sum of expectedCombination.py:30-35
    eng.bin_hist(X, N, S, want_hist=False, counts=counts)
    assert np.array_equal(_np(counts), 2 * h_ref.sum(axis=0))


@pytest.mark.parametrize("N,R", [(10, 333), (833, 257), (100, 64), (16, 5), (7, 1)])
@pytest.mark.parametrize("offset", [1, 3, 8])
def test_bin_hist_unaligned_packed_rows(eng, N, R, offset):
    x = synth_states(R, N, seed=N + R + offset, uniform=True)
    X = _unaligned_states(x, offset)
    assert X.stride(0) == N
    H, counts = eng.bin_hist(X, N, S)
    h_ref = onp.bin_hist(x, S)
    assert np.array_equal(eng.hist_to_numpy(H).astype(np.int64), h_ref)
    assert np.array_equal(_np(counts), h_ref.sum(axis=0))


@pytest.mark.parametrize("S_", [15, 25, 7, 31, 2])
def test_bin_hist_other_state_models(eng, S_):
    x = synth_states(777, 127, S=S_, seed=S_, uniform=True)
    X = eng.states_to_device(x)
    H, counts = eng.bin_hist(X, 127, S_)
    h_ref = onp.bin_hist(x, S_)
    assert np.array_equal(eng.hist_to_numpy(H).astype(np.int64), h_ref)
    assert np.array_equal(_np(counts), h_ref.sum(axis=0))


@pytest.mark.parametrize("S_,N", [(15, 127), (15, 833), (25, 200), (25, 1000), (18, 1000), (18, 1024), (18, 1025), (18, 1698), (18, 2500), (15, 4097)])
def test_bin_hist_fast_paths_of_other_models(eng, S_, N):
    """15- and 25-state ChromHMM models (reference data/state_metadata) and N in (896, 1024] take templated kernels."""
    x = synth_states(300, N, S=S_, seed=S_ + N, uniform=True)
    X = eng.states_to_device(x)
    H, counts = eng.bin_hist(X, N, S_)
    h_ref = onp.bin_hist(x, S_)
    assert np.array_equal(eng.hist_to_numpy(H).astype(np.int64), h_ref)
    assert np.array_equal(_np(counts), h_ref.sum(axis=0))
    q = onp.normalise(h_ref.sum(axis=0))
    o32, o64 = eng.score_s1(X, N, S_, torch.from_numpy(q).cuda(), want32=True, want64=True)
    np.testing.assert_allclose(_np(o64), onp.score_s1(x, q, S_), rtol=RTOL_TIGHT, atol=1e-15)


def test_bin_hist_ignores_invalid_states(eng):
    x = synth_states(100, 50, seed=5)
    x[3, 7] = -1      # a 0 in the input file
    x[9, 0] = 18      # state beyond the model
    X = eng.states_to_device(x)
    H, counts = eng.bin_hist(X, 50, S)
    h = eng.hist_to_numpy(H).astype(np.int64)
    assert h[3].sum() == 49 and h[9].sum() == 49 and h.sum() == 100 * 50 - 2
    assert int(_np(counts).sum()) == 100 * 50 - 2


def test_bin_hist_skewed_and_extreme(eng):
    # all-one-state bins (max count = N) and a maximum-width count
    x = np.full((64, 833), 17, dtype=np.int8)
    x[1] = 0
    x[2, ::2] = 5
    X = eng.states_to_device(x)
    H, counts = eng.bin_hist(X, 833, S)
    assert np.array_equal(eng.hist_to_numpy(H).astype(np.int64), onp.bin_hist(x, S))


def test_empty_input(eng):
    X = eng.alloc_states(0, 833)
    H, counts = eng.bin_hist(X, 833, S)
    assert H.shape == (0, S) and int(_np(counts).sum()) == 0


# ---------------------------------------------------------------------------------------------- S1
def _s1_check(eng, x, q_np, g64=None, g32=None):
    N = x.shape[1]
    X = eng.states_to_device(x)
    q = torch.from_numpy(q_np).cuda()
    o32, o64 = eng.score_s1(X, N, S, q, want32=True, want64=True)
    ref64 = onp.score_s1(x, q_np, S)
    np.testing.assert_allclose(_np(o64), ref64, rtol=RTOL_TIGHT, atol=1e-15)
    if g64 is not None:
        np.testing.assert_allclose(_np(o64), g64, rtol=RTOL_TIGHT, atol=1e-15)
    # float32 as stored: the float32 rounding of a value within 1e-11 of the reference's float64
    np.testing.assert_allclose(_np(o32), ref64.astype(np.float32), rtol=2e-7, atol=0)
    if g32 is not None:
        np.testing.assert_allclose(_np(o32), g32, rtol=2e-7, atol=0)
    # cached-histogram route gives the identical bits
    H, _ = eng.bin_hist(X, N, S)
    h32, h64 = eng.score_s1_from_binhist(H, N, S, q, want32=True, want64=True)
    assert torch.equal(h32, o32) and torch.equal(h64, o64)
    return _np(o32), _np(o64)


def test_s1_golden_real_slice(eng, golden_real):
    g = golden_real
    x = g["x"]
    X = eng.states_to_device(x)
    _, counts = eng.bin_hist(X, x.shape[1], S, want_hist=False)
    assert np.array_equal(_np(counts), g["s1_counts"])
    q = eng.normalise(counts)
    assert np.array_equal(_np(q), g["s1_exp"])          # bit-exact float32 exp_freq
    o32, o64 = _s1_check(eng, x, g["s1_exp"], g["s1_f64"], g["s1_f32"])
    print("S1 real slice: float32 bit-exact rows: %d / %d" % ((o32 == g["s1_f32"]).all(axis=1).sum(), o32.shape[0]))


def test_s1_golden_synth833(eng, golden_synth):
    g = golden_synth
    x = g["x"]
    X = eng.states_to_device(x)
    _, counts = eng.bin_hist(X, 833, S, want_hist=False)
    assert np.array_equal(_np(counts), g["s1_counts"])
    assert np.array_equal(_np(eng.normalise(counts)), g["s1_exp"])
    _s1_check(eng, x, g["s1_exp"], g["s1_f64"], g["s1_f32"])


def test_s1_edge_q_zero_and_all_states(eng, golden_edge):
    g = golden_edge
    X = eng.states_to_device(g["q0_probe"])
    q = torch.from_numpy(g["q0_exp"]).cuda()
    _, o64 = eng.score_s1(X, g["q0_probe"].shape[1], S, q, want32=False, want64=True)
    np.testing.assert_allclose(_np(o64), g["q0_probe_f64"], rtol=RTOL_TIGHT, atol=1e-15)
    assert _np(o64)[0, 3] == 0.0                         # q == 0 -> masked to 0, not inf/nan
    _s1_check(eng, g["all_x"], g["all_exp"], g["all_f64"], g["all_f32"])
    _s1_check(eng, g["n2_x"], g["n2_s1_exp"], g["n2_s1_f64"], g["n2_s1_f32"])


@pytest.mark.parametrize("N,R", [(379, 3000), (342, 1025), (1100, 300), (64, 4096)])
def test_s1_random_vs_oracle(eng, N, R):
    x = synth_states(R, N, seed=N)
    q = onp.normalise(onp.expected_s1(x, S))
    _s1_check(eng, x, q)


def test_s1_unaligned(eng):
    x = synth_states(501, 833, seed=11)
    q_np = onp.normalise(onp.expected_s1(x, S))
    X = _unaligned_states(x, 5)
    q = torch.from_numpy(q_np).cuda()
    o32, o64 = eng.score_s1(X, 833, S, q, want32=True, want64=True)
    np.testing.assert_allclose(_np(o64), onp.score_s1(x, q_np, S), rtol=RTOL_TIGHT, atol=1e-15)


def test_s1_meets_north_star_tolerance(eng, golden_synth):
    g = golden_synth
    X = eng.states_to_device(g["x"])
    q = torch.from_numpy(g["s1_exp"]).cuda()
    _, o64 = eng.score_s1(X, 833, S, q, want32=False, want64=True)
    np.testing.assert_allclose(_np(o64), g["s1_f64"], rtol=RTOL, atol=0)


# ---------------------------------------------------------------------------------------------- S2
def _s2_check(eng, x, q_np, g64=None, g32=None, perms=None, S=S):
    N = x.shape[1]
    X = eng.states_to_device(x)
    q = torch.from_numpy(np.ascontiguousarray(q_np).reshape(-1)).cuda()
    o32, o64 = eng.score_s2(X, N, S, q, perms=perms, want32=True, want64=True)
    ref64 = onp.score_s2(x, q_np, S, perms=perms)
    np.testing.assert_allclose(_np(o64), ref64, rtol=RTOL, atol=ATOL)
    if g64 is not None:
        np.testing.assert_allclose(_np(o64), g64, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(_np(o32), ref64.astype(np.float32), rtol=3e-7, atol=ATOL)
    if g32 is not None:
        np.testing.assert_allclose(_np(o32), g32, rtol=3e-7, atol=ATOL)
    err = np.abs(_np(o64) - ref64) / np.maximum(np.abs(ref64), 1e-300)
    return float(np.max(np.where(np.abs(ref64) > 1e-9, err, 0)))


def test_s2_counts_golden(eng, golden_real, golden_synth):
    for g in (golden_real, golden_synth):
        x = g["x"]
        X = eng.states_to_device(x)
        H, _ = eng.bin_hist(X, x.shape[1], S)
        c2 = eng.hist_s2_from_binhist(H, S)
        assert np.array_equal(_np(c2).reshape(S, S), g["s2_counts"])
        assert np.array_equal(_np(eng.normalise(c2)).reshape(S, S), g["s2_exp"])


def test_s2_score_golden(eng, golden_real, golden_synth):
    e1 = _s2_check(eng, golden_real["x"], golden_real["s2_exp"], golden_real["s2_f64"], golden_real["s2_f32"])
    e2 = _s2_check(eng, golden_synth["x"], golden_synth["s2_exp"], golden_synth["s2_f64"], golden_synth["s2_f32"])
    print("S2 max rel err vs oracle: real %.3g synth833 %.3g" % (e1, e2))


def test_s2_edge(eng, golden_edge):
    g = golden_edge
    _s2_check(eng, g["q0_x"], g["q0_s2_exp"], g["q0_s2_f64"], g["q0_s2_f32"])
    _s2_check(eng, g["n2_x"], g["n2_s2_exp"], g["n2_s2_f64"], g["n2_s2_f32"])


@pytest.mark.parametrize("N,R", [(379, 2000), (833, 700), (40, 3001)])
def test_s2_random_vs_oracle(eng, N, R):
    x = synth_states(R, N, seed=N + 1)
    c2 = onp.expected_s2(x, S)
    X = eng.states_to_device(x)
    H, _ = eng.bin_hist(X, N, S)
    assert np.array_equal(_np(eng.hist_s2_from_binhist(H, S)).reshape(S, S), c2)
    _s2_check(eng, x, onp.normalise(c2))


@pytest.mark.parametrize("S_,N,R", [(25, 2000, 300), (18, 3500, 200), (25, 1791, 130), (25, 1792, 130), (15, 4000, 100)])
def test_s2_score_wide_groups_float64_output(eng, S_, N, R):
    """The bin-per-lane score kernel stages 4 x 64 x S outputs in static LDS next to the dynamic log table; with float64
    outputs and N in the thousands the two no longer fit 64 KB together and the table must stay in memory (round-2 advisory:
    the launch would have failed and left the scores unwritten)."""
    x = synth_states(R, N, S=S_, seed=S_ * N, uniform=True)
    _s2_check(eng, x, onp.normalise(onp.expected_s2(x, S_)), S=S_)


# ---------------------------------------------------------------------------------------------- paired extras
@pytest.mark.parametrize("S_,N,R", [(2, 40, 1000), (10, 833, 777), (16, 130, 2049), (20, 833, 1031), (26, 64, 333), (31, 300, 1000), (15, 50, 31)])
def test_bin_hist_any_state_model(eng, S_, N, R):
    """State models between the instantiated sizes of the counting core (15, 18, 25, 31) run on the next larger one and
    store only their own columns; odd S with an odd number of rows in the last tile ends on a two-byte store."""
    x = synth_states(R, N, S=S_, seed=S_ + N, uniform=True)
    x[R // 3, N // 2] = S_                            # not a state of this model, though the counting core can decode it
    x[R // 2, 0] = -1
    X = eng.states_to_device(x)
    guard = torch.full((R * S_ + 64,), 0x5a5a, dtype=torch.int16, device="cuda")     # H with a canary behind it
    H = guard[: R * S_].view(R, S_)
    H, counts = eng.bin_hist(X, N, S_, H=H)
    want = onp.bin_hist(x, S_)
    assert np.array_equal(eng.hist_to_numpy(H), want)
    assert np.array_equal(_np(counts), want.sum(axis=0))
    assert (guard[R * S_:] == 0x5a5a).all()
    q = eng.normalise(counts)
    o32, o64 = eng.score_s1_from_binhist(H, N, S_, q, want32=True, want64=True)
    ref = onp.score_s1(x, onp.normalise(want.sum(axis=0)), S_)
    np.testing.assert_allclose(_np(o64), ref, rtol=1e-11, atol=0)
    np.testing.assert_allclose(_np(o32), ref.astype(np.float32), rtol=2e-7, atol=0)


@pytest.mark.parametrize("packed", [False, True])
def test_quiescent_wide_rows(eng, packed):
    """Rows of 379 + 342 bytes: all-quiescent rows, and rows with exactly one other state at the first byte, the last
    byte, either side of the 256-byte step and in group B only; padded and packed (unaligned rows) layouts."""
    R, NA, NB, q = 300, 379, 342, 17
    xa = np.full((R, NA), q, dtype=np.int8)
    xb = np.full((R, NB), q, dtype=np.int8)
    spots = [(10, "a", 0), (11, "a", NA - 1), (12, "a", 255), (13, "a", 256), (14, "b", 0), (15, "b", NB - 1), (16, "b", 300),
             (17, "a", 15), (18, "a", 16), (19, "b", 336)]
    for r, grp, c in spots:
        (xa if grp == "a" else xb)[r, c] = 3
    xa[100:200] = synth_states(100, NA, seed=5)
    if packed:
        XA, XB = torch.from_numpy(xa.copy()).cuda(), torch.from_numpy(xb.copy()).cuda()
    else:
        XA, XB = eng.states_to_device(xa), eng.states_to_device(xb)
    want = onp.quiescent_mask(xa, xb, q)
    assert want.sum() == R - len(spots) - 100
    assert np.array_equal(_np(eng.quiescent(XA, NA, XB, NB, q)).astype(bool), want)
    assert not _np(eng.quiescent(XA, NA, XB, NB, 3)).any()


@pytest.mark.parametrize("S_,R,hi", [(18, 1000, 834), (5, 129, 4096), (25, 300, 900), (31, 257, 4095), (18, 700, 65536), (30, 1, 70)])
def test_s2_counts_from_arbitrary_histograms(eng, S_, R, hi):
    """epg_hist_s2_from_binhist (k_s2_hist_wave) on histograms it is handed directly: every S up to 31 (one to three pair roles per
    thread), batches that end mid-way, and counts >= 4096, which leave the 32-bit dot-product path."""
    rng = np.random.default_rng(S_ * 1000 + R)
    h = rng.integers(0, hi, size=(R, S_)).astype(np.uint16)
    if hi > 4096:
        h[: R // 2] %= 4096                        # both paths within one call
    H = torch.from_numpy(h.view(np.int16)).cuda()
    c = eng.hist_s2_from_binhist(H, S_)
    h64 = h.astype(np.int64)
    want = h64.T @ h64 - np.diag(h64.sum(axis=0))
    assert np.array_equal(_np(c).reshape(S_, S_), want)
    c = eng.hist_s2_from_binhist(H, S_, counts=c)  # accumulates
    assert np.array_equal(_np(c).reshape(S_, S_), 2 * want)


def test_pair_finish_and_quiescent(eng, golden_pair):
    g = golden_pair
    for sal in (1, 2):
        p = "s%d_" % sal
        a = torch.from_numpy(g[p + "a"]).cuda()
        b = torch.from_numpy(g[p + "b"]).cuda()
        delta, _ = eng.pair_finish(a, b)
        assert np.array_equal(_np(delta), g[p + "delta"])
        na = torch.from_numpy(g[p + "na"]).cuda()
        nb = torch.from_numpy(g[p + "nb"]).cuda()
        _, dist = eng.pair_finish(na, nb)
        assert np.array_equal(_np(dist), g[p + "null_dist"])       # float32, numpy's summation order
    XA = eng.states_to_device(g["xa"])
    XB = eng.states_to_device(g["xb"])
    m = eng.quiescent(XA, 5, XB, 5, int(g["qstate"]))
    assert np.array_equal(_np(m).astype(bool), g["s1_quiescent"])
    assert not _np(eng.quiescent(XA, 5, XB, 5, -1)).any()


def test_paired_scores_golden(eng, golden_pair):
    g = golden_pair
    xa, xb = g["xa"], g["xb"]
    comb = np.concatenate([xa, xb], axis=1)
    XC = eng.states_to_device(comb)
    H, counts = eng.bin_hist(XC, 10, S)
    assert np.array_equal(_np(counts), g["s1_counts"])
    q1 = eng.normalise(counts)
    assert np.array_equal(_np(q1), g["s1_exp"])
    c2 = eng.hist_s2_from_binhist(H, S)
    assert np.array_equal(_np(c2).reshape(S, S), g["s2_counts"])
    q2 = eng.normalise(c2)
    for xs, key in ((xa, "a"), (xb, "b")):
        X = eng.states_to_device(xs)
        o32, _ = eng.score_s1(X, 5, S, q1)
        np.testing.assert_allclose(_np(o32), g["s1_" + key], rtol=2e-7, atol=0)
        o32, _ = eng.score_s2(X, 5, S, q2)
        np.testing.assert_allclose(_np(o32), g["s2_" + key], rtol=3e-7, atol=ATOL)


@pytest.mark.parametrize("NA,NB,ga,gb,R", [(379, 342, 379, 342, 5000), (5, 5, 5, 5, 2048), (40, 33, 20, 20, 3001), (7, 9, 9, 7, 64), (12, 12, 12, 12, 1),
                                           # wider groups: larger tables leave room for 7 / 5 waves' staging areas instead of 12
                                           (620, 600, 620, 600, 1500), (833, 700, 833, 700, 700), (500, 480, 100, 100, 1111)])
def test_paired_s1_in_one_pass_equals_the_separate_passes(eng, NA, NB, ga, gb, R):
    """epg_pair_scores_s1_from_binhist (scores of A, B and the null groups as table gathers, delta, null distance, STEP 4's
    reduction, one pass over the four histograms) against four epg_score_s1_from_binhist_table + two epg_pair_finish +
    epg_pair_metrics: bit for bit, for equal and unequal null widths (-g) and ragged row counts; and against the oracle."""
    from epilogos_amd.scores import s1ScoreTable
    xa, xb = synth_states(R, NA, seed=NA), synth_states(R, NB, seed=NB + 1)
    XA, XB = eng.states_to_device(xa), eng.states_to_device(xb)
    HA, cA = eng.bin_hist(XA, NA, S)
    HB, _ = eng.bin_hist(XB, NB, S, counts=cA)
    q = eng.normalise(cA).cpu().numpy()
    HnA, HnB = eng.null_hist_from_binhist(HA, HB, NA + NB, S, ga, gb, seed=3)
    tabs = {}
    for n in (NA, NB, ga, gb):
        if n not in tabs:
            tabs[n] = torch.from_numpy(s1ScoreTable(q, n)[1]).cuda()
    delta, null, dist, mdiff = eng.pair_scores_s1_from_binhist(HA, HB, HnA, HnB, S, NA, NB, ga, gb, tabs[NA], tabs[NB], tabs[ga], tabs[gb])
    sA, _ = eng.score_s1_from_binhist_table(HA, NA, S, T32=tabs[NA])
    sB, _ = eng.score_s1_from_binhist_table(HB, NB, S, T32=tabs[NB])
    nA, _ = eng.score_s1_from_binhist_table(HnA, ga, S, T32=tabs[ga])
    nB, _ = eng.score_s1_from_binhist_table(HnB, gb, S, T32=tabs[gb])
    d2, _ = eng.pair_finish(sA, sB, want_dist=False)
    _, n2 = eng.pair_finish(nA, nB)
    r2, m2 = eng.pair_metrics(d2, roundtrip=True)
    assert torch.equal(delta, d2) and torch.equal(null, n2) and torch.equal(dist, r2) and torch.equal(mdiff, m2)
    ref_a = onp.score_s1(xa, q, S).astype(np.float32)
    ref_b = onp.score_s1(xb, q, S).astype(np.float32)
    rd, _ = onp.pair_finish(ref_a, ref_b)
    assert np.array_equal(_np(delta), rd)
    wd, wx = onp.pair_metrics(rd, True)
    assert np.array_equal(_np(dist), wd) and np.array_equal(_np(mdiff), wx)


def test_paired_s1_in_one_pass_says_when_the_tables_do_not_fit(eng):
    """Groups so wide that the S1 tables leave no room for four waves' staging areas in a CU's LDS: EPG_ERR_UNSUPPORTED (-2), which
    backend._HipPairedSession answers with the separate passes."""
    from epilogos_amd.scores import s1ScoreTable
    NA, NB, R = 900, 880, 128
    xa, xb = synth_states(R, NA, seed=1), synth_states(R, NB, seed=2)
    HA, cA = eng.bin_hist(eng.states_to_device(xa), NA, S)
    HB, _ = eng.bin_hist(eng.states_to_device(xb), NB, S, counts=cA)
    q = eng.normalise(cA).cpu().numpy()
    HnA, HnB = eng.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=3)
    tA, tB = (torch.from_numpy(s1ScoreTable(q, n)[1]).cuda() for n in (NA, NB))
    with pytest.raises(eng.EpilogosHipError) as e:
        eng.pair_scores_s1_from_binhist(HA, HB, HnA, HnB, S, NA, NB, NA, NB, tA, tB, tA, tB)
    assert e.value.code == -2


@pytest.mark.parametrize("nparts,S_", [(3, 18), (30, 18), (5, 15), (4, 21)])
def test_paired_s1_several_parts_in_one_launch(eng, nparts, S_):
    """epg_pair_scores_s1_parts: the fused pass over SEVERAL parts (one per chromosome file in the command line) in one launch,
    quiescence masks included, against one epg_pair_scores_s1_from_binhist + epg_quiescent_from_binhist call per part -- bit for
    bit; parts of 1 row, of ragged sizes, an EMPTY part in the middle, more parts than fit one launch (24), a state count with
    and without a compile-time instantiation; quiescent state off (-1) gives all-zero masks."""
    from epilogos_amd.scores import s1ScoreTable
    NA, NB = 37, 29
    rng = np.random.default_rng(nparts)
    sizes = [int(v) for v in rng.integers(1, 700, size=nparts)]
    sizes[1] = 0
    sizes[-1] = 1
    R = sum(sizes)
    xa, xb = synth_states(R, NA, S=S_, seed=1), synth_states(R, NB, S=S_, seed=2)
    xa[5:9, :] = S_ - 1
    xb[5:8, :] = S_ - 1                                       # bins 5..7 are quiescent in both groups
    HA, cA = eng.bin_hist(eng.states_to_device(xa), NA, S_)
    HB, _ = eng.bin_hist(eng.states_to_device(xb), NB, S_, counts=cA)
    q = eng.normalise(cA).cpu().numpy()
    HnA, HnB = eng.null_hist_from_binhist(HA, HB, NA + NB, S_, NA, NB, seed=3)
    tA, tB = (torch.from_numpy(s1ScoreTable(q, n)[1]).cuda() for n in (NA, NB))
    # every part in an allocation of its own (16-byte aligned bases), like the session's parts
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    quads = [tuple(t[cuts[k]:cuts[k + 1]].clone() for t in (HA, HB, HnA, HnB)) for k in range(nparts)]
    for qstate in (S_ - 1, -1):
        res = eng.pair_scores_s1_parts(quads, S_, NA, NB, NA, NB, tA, tB, tA, tB, qstate=qstate)
        assert len(res) == nparts
        for k, (a, b, na, nb) in enumerate(quads):
            r = res[k]
            if sizes[k] == 0:
                assert r["delta"].shape == (0, S_) and r["quies"].numel() == 0
                continue
            d, n, rd, md = eng.pair_scores_s1_from_binhist(a, b, na, nb, S_, NA, NB, NA, NB, tA, tB, tA, tB)
            m = eng.quiescent_from_binhist(a, NA, b, NB, S_, qstate)
            assert torch.equal(r["delta"], d) and torch.equal(r["null"], n) and torch.equal(r["rdist"], rd) and torch.equal(r["mdiff"], md), k
            assert torch.equal(r["quies"], m), k
        allq = torch.cat([r["quies"] for r in res])
        assert int(allq.sum()) == (3 if qstate >= 0 else 0)
    # without masks
    res = eng.pair_scores_s1_parts(quads[:1], S_, NA, NB, NA, NB, tA, tB, tA, tB)
    assert res[0]["quies"] is None


@pytest.mark.parametrize("S_,R", [(15, 63), (15, 317), (21, 61), (25, 129), (15, 1)])
def test_paired_s1_one_pass_with_an_odd_number_of_counts_in_the_last_tile(eng, S_, R):
    """An odd state count x an odd number of rows in a wave's last tile: the staged histogram rows end on a two-byte tail, which
    the loader left out until round 4 (the last state's count of the last bin came from stale LDS).  The one-pass kernel against
    the separate passes, bit for bit."""
    from epilogos_amd.scores import s1ScoreTable
    NA, NB = 37, 29
    xa, xb = synth_states(R, NA, S=S_, seed=1), synth_states(R, NB, S=S_, seed=2)
    HA, cA = eng.bin_hist(eng.states_to_device(xa), NA, S_)
    HB, _ = eng.bin_hist(eng.states_to_device(xb), NB, S_, counts=cA)
    q = eng.normalise(cA).cpu().numpy()
    HnA, HnB = eng.null_hist_from_binhist(HA, HB, NA + NB, S_, NA, NB, seed=3)
    tA, tB = (torch.from_numpy(s1ScoreTable(q, n)[1]).cuda() for n in (NA, NB))
    for rep in range(3):                                      # (stale LDS differs from launch to launch)
        d, n, rd, md = eng.pair_scores_s1_from_binhist(HA, HB, HnA, HnB, S_, NA, NB, NA, NB, tA, tB, tA, tB)
        sA, _ = eng.score_s1_from_binhist_table(HA, NA, S_, T32=tA)
        sB, _ = eng.score_s1_from_binhist_table(HB, NB, S_, T32=tB)
        nA, _ = eng.score_s1_from_binhist_table(HnA, NA, S_, T32=tA)
        nB, _ = eng.score_s1_from_binhist_table(HnB, NB, S_, T32=tB)
        d2, _ = eng.pair_finish(sA, sB, want_dist=False)
        _, n2 = eng.pair_finish(nA, nB)
        r2, m2 = eng.pair_metrics(d2, roundtrip=True)
        assert torch.equal(d, d2) and torch.equal(n, n2) and torch.equal(rd, r2) and torch.equal(md, m2)

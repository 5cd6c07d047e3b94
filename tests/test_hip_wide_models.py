"""State models of more than 31 states (the reference takes any numStates, helpers.py:9-17; ChromHMM's full-stack model has 100):
the entry points hand them to the plain kernels of csrc/epg_wide.hip (whole-byte decode), the parser keeps file values up to
127.  Same oracle, same tolerances as the fast kernels: integers exact, float64 scores 1e-11 (S1) / 1e-6 rel 1e-12 abs (S2) /
1e-6 (S3)."""
import gzip

import numpy as np
import pytest

from epilogos_amd import _io
from oracle import oracle_np as onp
from tests.test_host_logic import write_tsv


@pytest.fixture(autouse=True)
def _narrow_again():
    yield
    _io.set_state_limit(31)


def test_parser_keeps_values_up_to_127_for_wide_models(tmp_path):
    """CPU: the native parser stores values outside 1..31 as "not a state" for the models the fast kernels serve and keeps
    1..127 once a wider model is announced (AVX-512 rows, the scalar fast path, three-digit values, the range report)."""
    rng = np.random.default_rng(2)
    x = rng.integers(0, 100, size=(300, 90)).astype(np.int8)
    x[5, 7] = 126
    f = tmp_path / "wide_chr1.txt.gz"
    write_tsv(f, x)
    _io.set_state_limit(18)
    st, _loc, rng_seen = _io.read_table(f, with_range=True)
    assert rng_seen == (1, 127)
    assert np.array_equal(st, np.where(x <= 30, x, -1))
    _io.set_state_limit(100)
    for threads in (0, 1):
        st, _loc, rng_seen = _io.read_table(f, with_range=True, threads=threads)
        assert rng_seen == (1, 127) and np.array_equal(st, x)
    (tmp_path / "big.txt").write_text("chr1\t0\t200\t128\t5\t300\n")
    st, _ = _io.read_table(tmp_path / "big.txt")
    assert st.tolist() == [[-1, 4, -1]]


@pytest.mark.gpu
@pytest.mark.parametrize("S,N,R", [(40, 37, 900), (100, 130, 401), (127, 70, 130), (32, 833, 65)])
def test_wide_s1_s2_against_oracle(S, N, R):
    import torch
    from epilogos_amd import engine
    engine.require_gpu()
    rng = np.random.default_rng(S * N)
    x = rng.integers(0, S, size=(R, N)).astype(np.int8)
    x[R // 2, N // 3] = -1                                                  # not a state
    if S < 127:
        x[R // 3, 0] = S                                                    # the first value outside the model
    X = engine.states_to_device(x)
    H, counts = engine.bin_hist(X, N, S)
    want_h = onp.bin_hist(x, S)
    assert np.array_equal(engine.hist_to_numpy(H), want_h)
    assert np.array_equal(counts.cpu().numpy(), want_h.sum(axis=0))
    xv = x.copy()
    xv[R // 2, N // 3] = 0
    if S < 127:
        xv[R // 3, 0] = 0
    Xv = engine.states_to_device(xv)
    Hv, cv = engine.bin_hist(Xv, N, S)
    q = engine.normalise(cv)
    assert np.array_equal(q.cpu().numpy(), onp.normalise(onp.expected_s1(xv, S)))
    # S1: device table, fused entry, host table (the reference's bits)
    o32, o64 = engine.score_s1_from_binhist(Hv, N, S, q, want32=True, want64=True)
    ref = onp.score_s1(xv, q.cpu().numpy(), S)
    np.testing.assert_allclose(o64.cpu().numpy(), ref, rtol=1e-11, atol=0)
    d32, _ = engine.score_s1(Xv, N, S, q)
    assert torch.equal(d32, o32)
    q2, c32, _ = engine.combine_score_s1(cv.clone(), Hv, N, S)
    assert torch.equal(q2, q) and torch.equal(c32, o32)
    from epilogos_amd.scores import s1ScoreTable
    t64, t32 = s1ScoreTable(q.cpu().numpy(), N)
    h32, h64 = engine.score_s1_from_binhist_table(Hv, N, S, T64=torch.from_numpy(t64).cuda(), T32=torch.from_numpy(t32).cuda())
    assert np.array_equal(h64.cpu().numpy(), ref) and np.array_equal(h32.cpu().numpy(), ref.astype(np.float32))
    # S2
    c2 = engine.hist_s2_from_binhist(Hv, S)
    want2 = onp.expected_s2(xv, S)
    assert np.array_equal(c2.cpu().numpy().reshape(S, S), want2)
    cp = engine.hist_s2_from_binhist_pair(Hv, Hv, S)
    assert np.array_equal(cp.cpu().numpy().reshape(S, S), onp.expected_s2(np.concatenate([xv, xv], axis=1), S))
    q2d = engine.normalise(c2)
    s32, s64 = engine.score_s2_from_binhist(Hv, N, S, q2d, want32=True, want64=True)
    ref2 = onp.score_s2(xv, q2d.cpu().numpy().reshape(S, S), S)
    np.testing.assert_allclose(s64.cpu().numpy(), ref2, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(s32.cpu().numpy(), ref2.astype(np.float32), rtol=3e-7, atol=1e-12)
    x32, _ = engine.score_s2(Xv, N, S, q2d)
    assert torch.equal(x32, s32)


@pytest.mark.gpu
@pytest.mark.parametrize("S,N,R", [(40, 12, 300), (100, 9, 120)])
def test_wide_s3_against_oracle(S, N, R):
    import torch
    from epilogos_amd import engine
    engine.require_gpu()
    rng = np.random.default_rng(S + N)
    x = rng.integers(0, S, size=(R, N)).astype(np.int8)
    X = engine.states_to_device(x)
    c3 = engine.hist_s3(X, N, S)
    want = onp.expected_s3(x, S)
    assert np.array_equal(c3.cpu().numpy().reshape(N, N, S, S), want)
    c3b = engine.hist_s3(X, N, S, counts=c3.clone(), use_workspace=False)
    assert torch.equal(c3b, 2 * c3)
    q = engine.normalise(c3)
    assert np.array_equal(q.cpu().numpy().reshape(N, N, S, S), onp.normalise(want))
    o32, o64 = engine.score_s3(X, N, S, q, want32=True, want64=True)
    o32b, _ = engine.score_s3(X, N, S, q, want32=True, want64=False)
    assert torch.equal(o32, o32b) and torch.equal(o32, o64.to(torch.float32))
    np.testing.assert_allclose(o64.cpu().numpy(), onp.score_s3_f64(x, q.cpu().numpy().reshape(N, N, S, S), S), rtol=1e-6, atol=1e-9)
    xd = x.copy()
    xd[3, 2] = -1                                                           # a byte that is not a state adds nothing
    cd = engine.hist_s3(engine.states_to_device(xd), N, S)
    assert int(cd.sum(dtype=torch.int64)) == R * N * (N - 1) - 2 * (N - 1)


@pytest.mark.gpu
def test_wide_paired_extras():
    import torch
    from epilogos_amd import engine
    engine.require_gpu()
    S, NA, NB, R = 100, 60, 45, 3000
    rng = np.random.default_rng(9)
    p = rng.dirichlet(np.full(S, 0.2))
    xa, xb = rng.choice(S, size=(R, NA), p=p).astype(np.int8), rng.choice(S, size=(R, NB), p=p).astype(np.int8)
    xa[10:40] = S - 1
    xb[10:30] = S - 1
    XA, XB = engine.states_to_device(xa), engine.states_to_device(xb)
    HA, cA = engine.bin_hist(XA, NA, S)
    HB, _ = engine.bin_hist(XB, NB, S, counts=cA)
    q = engine.normalise(cA)
    assert np.array_equal(q.cpu().numpy(), onp.normalise(onp.expected_s1(np.concatenate([xa, xb], axis=1), S)))
    m = engine.quiescent_from_binhist(HA, NA, HB, NB, S, S - 1)
    want_m = onp.quiescent_mask(xa, xb, S - 1)
    assert np.array_equal(m.cpu().numpy().astype(bool), want_m) and want_m.sum() >= 20
    assert np.array_equal(engine.quiescent(XA, NA, XB, NB, S - 1).cpu().numpy().astype(bool), want_m)
    # null groups: every row keeps its combined histogram's total and the group sizes; means follow the hypergeometric law
    OA, OB = engine.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=5)
    OA2, _ = engine.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=5)
    assert torch.equal(OA, OA2)
    oa, ob = engine.hist_to_numpy(OA).astype(np.int64), engine.hist_to_numpy(OB).astype(np.int64)
    h = engine.hist_to_numpy(HA).astype(np.int64) + engine.hist_to_numpy(HB)
    assert np.array_equal(oa + ob, h) and (oa.sum(axis=1) == NA).all() and (ob.sum(axis=1) == NB).all()
    tot = h.sum(axis=0)
    expect = tot * NA / (NA + NB)
    big = expect > 200
    assert big.sum() >= 3
    sd = np.sqrt(tot * (NA / (NA + NB)) * (NB / (NA + NB)))                  # (upper bound of) the hypergeometric spread of a column sum
    assert (np.abs(oa.sum(axis=0) - expect)[big] < 5 * sd[big]).all()
    # deltas and STEP 4's reduction: numpy's float32 pairwise order / ascending-state sums, bit for bit
    sa, _ = engine.score_s1_from_binhist(HA, NA, S, q)
    sb, _ = engine.score_s1_from_binhist(HB, NB, S, q)
    delta, dist = engine.pair_finish(sa, sb)
    rd, rdist = onp.pair_finish(sa.cpu().numpy(), sb.cpu().numpy())
    assert np.array_equal(delta.cpu().numpy(), rd) and np.array_equal(dist.cpu().numpy(), rdist)
    md, mx = engine.pair_metrics(delta, roundtrip=True)
    wd, wx = onp.pair_metrics(rd, True)
    assert np.array_equal(md.cpu().numpy(), wd) and np.array_equal(mx.cpu().numpy(), wx)


@pytest.mark.gpu
def test_cli_with_a_40_state_model(tmp_path):
    """The command line on a 40-state model, single S1 and S2 over two files: parser (values up to 40), wide kernels, writer, STEP 4;
    scores against the oracle, and a file holding state 41 is refused before any GPU work."""
    from click.testing import CliRunner
    from epilogos_amd.run import main
    S, N = 40, 23
    rng = np.random.default_rng(4)
    x = rng.choice(S, size=(2500, N), p=rng.dirichlet(np.full(S, 0.3))).astype(np.int8)
    ind = tmp_path / "in"
    ind.mkdir()
    write_tsv(ind / "m_chr1.txt.gz", x[:1500], chrom="chr1")
    write_tsv(ind / "m_chr2.txt.gz", x[1500:], chrom="chr2")
    meta = tmp_path / "metadata.tsv"
    meta.write_text("zero_index\tone_index\tshort_name\n" + "".join("%d\t%d\tS%d\n" % (i, i + 1, i + 1) for i in range(S)))
    for sal in (1, 2):
        out = tmp_path / ("out%d" % sal)
        res = CliRunner().invoke(main, ["-l", "-i", str(ind), "-j", str(meta), "-o", str(out), "-s", str(sal), "-f", "t", "-w", "10"], catch_exceptions=False)
        assert res.exit_code == 0, res.output
        got = []
        for name in ("scores_t_m_chr1.txt.gz", "scores_t_m_chr2.txt.gz"):
            with gzip.open(out / name, "rt") as fh:
                got += [[float(v) for v in line.split("\t")[3:]] for line in fh]
        got = np.array(got)
        if sal == 1:
            ref = onp.score_s1(x, onp.normalise(onp.expected_s1(x, S)), S).astype(np.float32)
        else:
            ref = onp.score_s2(x, onp.normalise(onp.expected_s2(x, S)), S).astype(np.float32)
        np.testing.assert_allclose(got, ref, atol=1.01e-5)
        assert (out / "regionsOfInterest_t.txt").exists()
    bad = x[:50].astype(np.int64).copy()
    bad[3, 3] = S                                                            # file value 41
    bdir = tmp_path / "bad"
    bdir.mkdir()
    write_tsv(bdir / "m_chr1.txt", bad)
    with pytest.raises(ValueError):
        CliRunner().invoke(main, ["-l", "-i", str(bdir), "-j", str(meta), "-o", str(tmp_path / "ob")], catch_exceptions=False)

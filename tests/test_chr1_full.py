"""The whole chr1 example (BASELINE config 1: 1 246 253 bins x 10 biosamples, built by the reference's preprocessing
script from its bundled ChromHMM calls) against a digest of the REAL reference's end-to-end S1 run
(tests/golden/make_golden_chr1.py): counts, exp_freq, every 997th score row, float64 column sums, and the SHA-256 of the
scores text.  The CPU test pins the oracle (and the native writer) on all 1.2 M bins; the GPU test runs the HIP path on
all of them."""
import hashlib
from pathlib import Path

import numpy as np
import pytest

from epilogos_amd import _io
from oracle import oracle_np as onp

S = 18
GOLD = Path(__file__).parent / "golden" / "chr1_full.npz"


@pytest.fixture(scope="module")
def g():
    return dict(np.load(GOLD))


def _locations(g, R):
    start = int(g["start0"]) + 200 * np.arange(R, dtype=np.int64)
    blob = "".join("chr1\t%d\t%d\n" % (s, s + 200) for s in start).encode()
    off = np.zeros(R + 1, dtype=np.int64)
    np.cumsum([len(l) + 1 for l in blob.decode().split("\n")[:-1]], out=off[1:])
    return _io.Locations(np.frombuffer(blob, dtype=np.uint8).copy(), off)


def _text(tmp_path, loc, scores32):
    import gzip
    out = tmp_path / "scores.txt.gz"
    _io.write_scores(out, loc, scores32)
    with gzip.open(out, "rb") as fh:
        return fh.read()


def test_oracle_and_writer_match_reference_on_full_chr1(g, tmp_path):
    x = g["x"]
    R = x.shape[0]
    counts = onp.expected_s1(x, S)
    assert np.array_equal(counts, g["counts"])
    q = onp.normalise(counts)
    assert np.array_equal(q, g["exp"])
    s32 = onp.score_s1(x, q, S).astype(np.float32)
    assert np.array_equal(s32[::997], g["rows_997"])
    assert np.array_equal(s32.astype(np.float64).sum(axis=0), g["colsum_f64"])
    assert np.float32(np.abs(s32).max()) == g["absmax"]
    text = _text(tmp_path, _locations(g, R), s32)
    assert len(text) == int(g["text_bytes"])
    assert text.split(b"\n")[0] == g["first_line"].tobytes() and text.split(b"\n")[-2] == g["last_line"].tobytes()
    assert hashlib.sha256(text).digest() == g["text_sha256"].tobytes()      # 1.2 M lines, byte for byte the reference's file


def test_oracle_s2_and_paired_match_reference_on_full_chr1(g, tmp_path):
    x = g["x"]
    R = x.shape[0]
    loc = _locations(g, R)
    c2 = onp.expected_s2(x, S)
    assert np.array_equal(c2, g["s2_counts"])
    q2 = onp.normalise(c2)
    assert np.array_equal(q2, g["s2_exp"])
    s2 = np.concatenate([onp.score_s2(x[lo:lo + 100000], q2, S) for lo in range(0, R, 100000)]).astype(np.float32)
    assert np.array_equal(s2[::997], g["s2_rows_997"])
    assert np.array_equal(s2.astype(np.float64).sum(axis=0), g["s2_colsum_f64"])
    text = _text(tmp_path, loc, s2)
    assert len(text) == int(g["s2_text_bytes"]) and hashlib.sha256(text).digest() == g["s2_text_sha256"].tobytes()
    # paired S1: biosamples 0-4 against 5-9, background over all ten
    xa, xb = x[:, :5], x[:, 5:]
    cp = onp.expected_s1(x, S)
    assert np.array_equal(cp, g["pair_counts"])
    qp = onp.normalise(cp)
    assert np.array_equal(qp, g["pair_exp"])
    delta, _ = onp.pair_finish(onp.score_s1(xa, qp, S).astype(np.float32), onp.score_s1(xb, qp, S).astype(np.float32))
    text = _text(tmp_path, loc, delta)
    assert len(text) == int(g["pair_text_bytes"]) and hashlib.sha256(text).digest() == g["pair_text_sha256"].tobytes()
    assert int(onp.quiescent_mask(xa, xb, S - 1).sum()) == int(g["pair_quiescent_count"])


def test_oracle_s3_matches_reference_on_full_chr1(g):
    x = g["x"]
    c3 = onp.expected_s3(x, S)
    assert np.array_equal(c3, g["s3_counts"]) and c3.dtype == np.int32
    q3 = onp.normalise(c3)
    assert np.array_equal(q3, g["s3_exp"])
    rows = np.arange(0, x.shape[0], 997)
    s3 = onp.score_s3_f64(x[rows], q3, S)
    # the reference adds 90 float32 terms per bin one after the other; the oracle sums the same float32 table in float64
    np.testing.assert_allclose(s3, g["s3_rows_997"], rtol=1e-4, atol=5e-6)


@pytest.mark.gpu
def test_hip_s3_on_full_chr1(g):
    import torch
    from epilogos_amd import engine
    engine.require_gpu()
    x = g["x"]
    R, N = x.shape
    X = engine.states_to_device(x)
    c3 = engine.hist_s3(X, N, S)
    assert np.array_equal(c3.cpu().numpy().reshape(N, N, S, S), g["s3_counts"])
    q3 = engine.normalise(c3)
    assert np.array_equal(q3.cpu().numpy().reshape(N, N, S, S), g["s3_exp"])
    o32, o64 = engine.score_s3(X, N, S, q3, want32=True, want64=True)
    got = o32.cpu().numpy()
    np.testing.assert_allclose(got[::997], g["s3_rows_997"], rtol=1e-4, atol=5e-6)       # the reference's own rows
    np.testing.assert_allclose(got.astype(np.float64).sum(axis=0), g["s3_colsum_f64"], rtol=1e-5)
    rows = np.arange(0, R, 997)
    np.testing.assert_allclose(o64.cpu().numpy()[rows], onp.score_s3_f64(x[rows], g["s3_exp"], S), rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
def test_hip_s2_and_paired_on_full_chr1(g):
    import torch
    from epilogos_amd import engine
    engine.require_gpu()
    x = g["x"]
    R, N = x.shape
    X = engine.states_to_device(x)
    H, _ = engine.bin_hist(X, N, S)
    c2 = engine.hist_s2_from_binhist(H, S)
    assert np.array_equal(c2.cpu().numpy().reshape(S, S), g["s2_counts"])
    q2 = engine.normalise(c2)
    assert np.array_equal(q2.cpu().numpy().reshape(S, S), g["s2_exp"])
    o32, o64 = engine.score_s2_from_binhist(H, N, S, q2, want32=True, want64=True)
    got = o32.cpu().numpy()
    np.testing.assert_allclose(got[::997], g["s2_rows_997"], rtol=1e-6, atol=1e-12)       # the reference's own rows
    np.testing.assert_allclose(got.astype(np.float64).sum(axis=0), g["s2_colsum_f64"], rtol=1e-9)
    for lo in (0, 600000, R - 50000):                                                    # float64 against the oracle
        ref = onp.score_s2(x[lo:lo + 50000], g["s2_exp"], S)
        np.testing.assert_allclose(o64.cpu().numpy()[lo:lo + 50000], ref, rtol=1e-6, atol=1e-12)
    # paired
    XA, XB = engine.states_to_device(x[:, :5]), engine.states_to_device(x[:, 5:])
    qp = torch.from_numpy(g["pair_exp"]).cuda()
    sa, _ = engine.score_s1(XA, 5, S, qp)
    sb, _ = engine.score_s1(XB, 5, S, qp)
    delta, _ = engine.pair_finish(sa, sb)
    ref_delta, _ = onp.pair_finish(onp.score_s1(x[:, :5], g["pair_exp"], S).astype(np.float32),
                                   onp.score_s1(x[:, 5:], g["pair_exp"], S).astype(np.float32))
    np.testing.assert_allclose(delta.cpu().numpy(), ref_delta, rtol=0, atol=2e-7)
    m = engine.quiescent(XA, 5, XB, 5, S - 1)
    assert int(m.sum().item()) == int(g["pair_quiescent_count"])
    # the command line's route (round 3): tables from the host, all of it in one pass over the histograms -- the reference's
    # float32 deltas bit for bit, on all 1 246 253 bins
    from epilogos_amd.scores import s1ScoreTable
    HA, _ = engine.bin_hist(XA, 5, S, want_counts=False)
    HB, _ = engine.bin_hist(XB, 5, S, want_counts=False)
    t5 = torch.from_numpy(s1ScoreTable(g["pair_exp"], 5)[1]).cuda()
    HnA, HnB = engine.null_hist_from_binhist(HA, HB, 10, S, 5, 5, seed=1)
    d1, _null, rdist, mdiff = engine.pair_scores_s1_from_binhist(HA, HB, HnA, HnB, S, 5, 5, 5, 5, t5, t5, t5, t5)
    assert np.array_equal(d1.cpu().numpy(), ref_delta)
    wd, wx = onp.pair_metrics(ref_delta, True)
    assert np.array_equal(rdist.cpu().numpy(), wd) and np.array_equal(mdiff.cpu().numpy(), wx)


@pytest.mark.gpu
def test_hip_on_full_chr1(g, tmp_path):
    import torch
    from epilogos_amd import engine
    engine.require_gpu()
    x = g["x"]
    R, N = x.shape
    X = engine.states_to_device(x)
    H, counts = engine.bin_hist(X, N, S)
    assert np.array_equal(counts.cpu().numpy(), g["counts"])
    q = engine.normalise(counts)
    assert np.array_equal(q.cpu().numpy(), g["exp"])
    o32, o64 = engine.score_s1_from_binhist(H, N, S, q, want32=True, want64=True)
    d32, _ = engine.score_s1(X, N, S, q)
    assert torch.equal(o32, d32)
    ref64 = onp.score_s1(x, g["exp"], S)
    np.testing.assert_allclose(o64.cpu().numpy(), ref64, rtol=1e-11, atol=0)          # north star asks for 1e-6
    got = o32.cpu().numpy()
    ref32 = ref64.astype(np.float32)
    np.testing.assert_allclose(got, ref32, rtol=2e-7, atol=0)
    assert (got != ref32).mean() < 1e-3                                               # float32 stores differ at rounding ties only
    np.testing.assert_allclose(got[::997], g["rows_997"], rtol=2e-7, atol=0)          # the reference's own rows
    np.testing.assert_allclose(got.astype(np.float64).sum(axis=0), g["colsum_f64"], rtol=1e-9)
    # text: same lines as the reference's file except where a float32 differs in its last bit
    loc = _locations(g, R)
    mine, theirs = _text(tmp_path, loc, got).split(b"\n"), _text(tmp_path, loc, ref32).split(b"\n")
    assert len(mine) == len(theirs) and hashlib.sha256(b"\n".join(theirs)).digest() == g["text_sha256"].tobytes()
    same = sum(a == b for a, b in zip(mine, theirs))
    assert same >= 0.999 * len(theirs)
    # the command line's route: the score table comes from the host (scores.s1ScoreTable, numpy's log2 like the reference):
    # every float32 equals the reference's and the text IS the reference's file, all 1 246 253 lines
    from epilogos_amd.scores import s1ScoreTable
    t64, t32 = s1ScoreTable(g["exp"], N)
    h32, h64 = engine.score_s1_from_binhist_table(H, N, S, T64=torch.from_numpy(t64).cuda(), T32=torch.from_numpy(t32).cuda())
    assert np.array_equal(h64.cpu().numpy(), ref64) and np.array_equal(h32.cpu().numpy(), ref32)
    assert np.array_equal(h32.cpu().numpy()[::997], g["rows_997"])
    assert hashlib.sha256(_text(tmp_path, loc, h32.cpu().numpy())).digest() == g["text_sha256"].tobytes()
    from epilogos_amd import backend
    sc = backend.HipBackend().scores(x, S, 1, g["exp"])                                 # the stage driver's call (scores.py)
    assert hashlib.sha256(_text(tmp_path, loc, sc)).digest() == g["text_sha256"].tobytes()

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

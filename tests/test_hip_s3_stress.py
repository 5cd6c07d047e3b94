"""Random shapes through both S3 score kernels -- k_s3_score_bl (biosample lanes, the default for S <= 20) and k_s3_score
(EPG_S3_SCORE=bins) --: widths around the multiples of 32 (chunks of biosample lanes), bin counts around the multiples of 48
and 1440 (half-wave and workgroup slices), 2 to 20 states, skewed state distributions, a q with zero (masked) entries, bytes
that are not states.  The two kernels must agree to 1e-6, each run must repeat bit for bit, and the small shapes must match the
float64 oracle (scores.py:455-506 restated)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu


def test_s3_score_kernels_random_shapes(monkeypatch):
    from epilogos_amd import engine
    engine.require_gpu()
    rng = np.random.default_rng(5)
    checked_oracle = 0
    for case in range(40):
        S = int(rng.integers(2, 21))
        N = int(rng.choice([2, 3, 31, 32, 33, 63, 64, 65, 95, 97, 129, 200, int(rng.integers(2, 260))]))
        R = int(rng.choice([1, 47, 48, 95, 96, 97, 1439, 1440, 1441, 2881, int(rng.integers(1, 4000))]))
        if case < 8:                                                       # oracle-sized
            N, R = int(rng.integers(2, 40)), int(rng.integers(1, 150))
        x = rng.choice(S, size=(R, N), p=rng.dirichlet(np.full(S, 0.3))).astype(np.int8)
        small = case < 8                                                   # the oracle scores valid states only
        if not small and R > 3 and N > 2:
            x[rng.integers(0, R), rng.integers(0, N)] = -1
            x[rng.integers(0, R), rng.integers(0, N)] = 31
        q = rng.random((N, N, S, S)).astype(np.float32) ** 3
        q[rng.random(q.shape) < 0.05] = 0.0
        q /= q.sum()
        X = engine.states_to_device(x)
        qd = torch.from_numpy(q.reshape(-1)).cuda()
        monkeypatch.delenv("EPG_S3_SCORE", raising=False)
        a32, a64 = engine.score_s3(X, N, S, qd, want32=True, want64=True)
        b32, _ = engine.score_s3(X, N, S, qd, want32=True, want64=False)
        assert torch.equal(a32, b32), (N, S, R)
        monkeypatch.setenv("EPG_S3_SCORE", "bins")
        _, c64 = engine.score_s3(X, N, S, qd, want32=False, want64=True)
        monkeypatch.delenv("EPG_S3_SCORE")
        np.testing.assert_allclose(a64.cpu().numpy(), c64.cpu().numpy(), rtol=1e-6, atol=1e-9, err_msg=str((N, S, R)))
        if small:
            np.testing.assert_allclose(a64.cpu().numpy(), onp.score_s3_f64(x, q, S), rtol=2e-6, atol=1e-9, err_msg=str((N, S, R)))
            checked_oracle += 1
    assert checked_oracle == 8


def test_s3_expected_kernels_random_shapes(monkeypatch):
    """Random shapes through the expected-count paths: the matrix-core contraction over all S states (in its three schedules:
    ring of three, ring of four, ping-pong), the reduced one (S - 1 states per biosample, the last state's cells re-derived from
    the marginals; EPG_S3_REDUCED=1 forces it below its size threshold) and the LDS-counter kernel -- with and without bytes that are not states (which must switch the reduced
    path off on the device), and accumulating into non-zero counts.  expected.py:183-200."""
    from epilogos_amd import engine
    engine.require_gpu()
    rng = np.random.default_rng(11)
    for case in range(30):
        S = int(rng.integers(2, 31))
        N = int(rng.choice([2, 3, 5, 31, 32, 33, 64, 97, int(rng.integers(2, 140))]))
        R = int(rng.choice([1, 63, 64, 65, 511, 512, 513, 2049, int(rng.integers(1, 3000))]))
        x = rng.choice(S, size=(R, N), p=rng.dirichlet(np.full(S, 0.5))).astype(np.int8)
        dirty = case % 3 == 0 and R > 2
        if dirty:
            x[rng.integers(0, R), rng.integers(0, N)] = -1
            x[rng.integers(0, R), rng.integers(0, N)] = S          # the first value that is not a state
        X = engine.states_to_device(x)
        monkeypatch.setenv("EPG_S3_REDUCED", "1")
        red = engine.hist_s3(X, N, S)
        red2 = engine.hist_s3(X, N, S, counts=red.clone())
        monkeypatch.setenv("EPG_S3_REDUCED", "0")
        full = engine.hist_s3(X, N, S)
        monkeypatch.delenv("EPG_S3_REDUCED")
        lds = engine.hist_s3(X, N, S, use_workspace=False)
        assert torch.equal(red, full) and torch.equal(full, lds), (N, S, R, dirty)
        # the other schedules of the contraction kernel: a ring of four stages, the ping-pong of the SIMD partners
        monkeypatch.setenv("EPG_S3_RING", "4")
        ring4 = engine.hist_s3(X, N, S)
        monkeypatch.delenv("EPG_S3_RING")
        monkeypatch.setenv("EPG_S3_SYRK", "pp")
        monkeypatch.setenv("EPG_S3_REDUCED", "1" if case % 2 else "0")
        pp = engine.hist_s3(X, N, S)
        monkeypatch.delenv("EPG_S3_SYRK")
        monkeypatch.delenv("EPG_S3_REDUCED")
        assert torch.equal(ring4, full) and torch.equal(pp, full), (N, S, R, dirty)
        assert torch.equal(red2, 2 * full), (N, S, R, dirty)
        if not dirty and N * N * S * S * R < 4e8:                  # oracle-sized and clean: the reference's own counts
            assert np.array_equal(full.cpu().numpy().reshape(N, N, S, S), onp.expected_s3(x, S)), (N, S, R)


def test_s3_modal_state_kernel_returns_the_dense_kernels_bits(monkeypatch):
    """EPG_S3_SCORE=sparse (epg_s3_sparse.hip: per-biosample base table + gathers only for biosamples off the modal state,
    accumulators addressed through the VGPR index mode) against the dense biosample-lane kernel: the same fixed-point unit,
    so float64 and float32 scores are IDENTICAL -- chunks of 32 biosamples with ragged ends, slices of 1440 bins with ragged
    ends, 2 to 19 states (every slab size of its three-buffer ring, incl. the third buffer's offset bias), a q with zeros, and
    a matrix with bytes that are not states (handed to the dense kernel).  scores.py:455-506."""
    from epilogos_amd import engine
    engine.require_gpu()
    rng = np.random.default_rng(3)
    shapes = [(18, 33, 200), (18, 64, 1440), (18, 65, 1441), (18, 129, 4000), (15, 40, 300), (19, 97, 2500), (2, 5, 77), (7, 31, 129),
              (18, 200, 2881), (11, 32, 128), (16, 70, 1500), (17, 50, 500), (3, 9, 50), (10, 100, 3000), (13, 35, 97)]
    for n, (S, N, R) in enumerate(shapes):
        p = rng.dirichlet(np.full(S, 0.3))
        p[int(rng.integers(0, S))] += 2.0
        p /= p.sum()
        x = rng.choice(S, size=(R, N), p=p).astype(np.int8)
        if n % 4 == 3:
            x[rng.integers(0, R), rng.integers(0, N)] = -1
            x[rng.integers(0, R), rng.integers(0, N)] = 31
        q = rng.random((N, N, S, S)).astype(np.float32) ** 3
        q[rng.random(q.shape) < 0.05] = 0.0
        q /= q.sum()
        X = engine.states_to_device(x)
        qd = torch.from_numpy(q.reshape(-1)).cuda()
        monkeypatch.setenv("EPG_S3_SCORE", "lanes")
        d32, d64 = engine.score_s3(X, N, S, qd, want32=True, want64=True)
        monkeypatch.setenv("EPG_S3_SCORE", "sparse")
        s32, s64 = engine.score_s3(X, N, S, qd, want32=True, want64=True)
        s32b, _ = engine.score_s3(X, N, S, qd, want32=True, want64=False)
        monkeypatch.delenv("EPG_S3_SCORE")
        assert torch.equal(d64, s64) and torch.equal(d32, s32) and torch.equal(s32, s32b), (S, N, R)
        assert bool(torch.isfinite(s64).all())
        if R * N * N < 3e6 and n % 4 != 3:
            np.testing.assert_allclose(s64.cpu().numpy(), onp.score_s3_f64(x, q, S), rtol=2e-6, atol=1e-9, err_msg=str((S, N, R)))

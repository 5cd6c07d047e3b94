"""Random shapes through both S3 score kernels -- k_s3_score_bl (biosample lanes, the default for S <= 20) and k_s3_score
(epg_test_force(1, 1)) --: widths around the multiples of 32 (chunks of biosample lanes), bin counts around the multiples of 48
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
        a32, a64 = engine.score_s3(X, N, S, qd, want32=True, want64=True)
        b32, _ = engine.score_s3(X, N, S, qd, want32=True, want64=False)
        assert torch.equal(a32, b32), (N, S, R)
        engine._abi.call("epg_test_force", 1, 1)                          # the bin-per-lane kernel (the path for S > 21)
        try:
            _, c64 = engine.score_s3(X, N, S, qd, want32=False, want64=True)
        finally:
            engine._abi.call("epg_test_force", 1, 0)
        np.testing.assert_allclose(a64.cpu().numpy(), c64.cpu().numpy(), rtol=1e-6, atol=1e-9, err_msg=str((N, S, R)))
        if small:
            np.testing.assert_allclose(a64.cpu().numpy(), onp.score_s3_f64(x, q, S), rtol=1e-6, atol=1e-9, err_msg=str((N, S, R)))
            checked_oracle += 1
    assert checked_oracle == 8


def test_s3_expected_kernels_random_shapes(monkeypatch):
    """Random shapes through the expected-count paths: the matrix-core contraction over all S states, the reduced one (S - 1
    states per biosample, the last state's cells re-derived from the marginals; epg_test_force(2, 2) runs it below its size
    threshold) and the LDS-counter kernel -- with and without bytes that are not states (which must switch the reduced path off
    on the device), and accumulating into non-zero counts.  expected.py:183-200."""
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
        try:
            engine._abi.call("epg_test_force", 2, 2)
            red = engine.hist_s3(X, N, S)
            red2 = engine.hist_s3(X, N, S, counts=red.clone())
            engine._abi.call("epg_test_force", 2, 1)
            full = engine.hist_s3(X, N, S)
            engine._abi.call("epg_test_force", 2, 0)
            engine._abi.call("epg_test_force", 3, 1)                      # the LDS-counter kernel although there is a workspace
            lds_forced = engine.hist_s3(X, N, S)
        finally:
            engine._abi.call("epg_test_force", 2, 0)
            engine._abi.call("epg_test_force", 3, 0)
        lds = engine.hist_s3(X, N, S, use_workspace=False)
        assert torch.equal(red, full) and torch.equal(full, lds) and torch.equal(lds, lds_forced), (N, S, R, dirty)
        assert torch.equal(red2, 2 * full), (N, S, R, dirty)
        if not dirty and N * N * S * S * R < 4e8:                  # oracle-sized and clean: the reference's own counts
            assert np.array_equal(full.cpu().numpy().reshape(N, N, S, S), onp.expected_s3(x, S)), (N, S, R)


@pytest.mark.parametrize("dirty", [False, True])
def test_s3_expected_several_chunks(dirty):
    """A call whose workspace holds less than the call's whole one-hot operand is contracted chunk by chunk (epg_s3_gemm.hip
    hist_s3_gemm_run; the genome is seven to eight chunks of 2 M bins): here two to three chunks of ~50 K bins -- reduced and full
    contraction, with and without a byte that is not a state (which gates the reduced launches off on the device), accumulating
    into non-zero counts -- against the LDS-counter kernel, which has no chunks and no operand.  expected.py:183-200."""
    from epilogos_amd import engine
    engine.require_gpu()
    N, S, R = 40, 18, 100_000
    rng = np.random.default_rng(77)
    x = rng.choice(S, size=(R, N), p=rng.dirichlet(np.full(S, 0.5))).astype(np.int8)
    if dirty:
        x[R - 7, 3] = -1
        x[12345, 11] = S
    X = engine.states_to_device(x)
    want = engine.hist_s3(X, N, S, use_workspace=False)
    # workspace: transposed matrix + task list + reduced count array (generous bounds) + room for ~2.6 chunks of 20 K bins
    Rp = (R + 511) // 512 * 512
    NT = (N * S + 383) // 384 * 384 // 32                                  # operand tiles per 64 bins (1 KiB each)
    fixed = N * Rp + (1 << 16)
    reduced = N * N * (S - 1) * (S - 1) * 4 + N * 32 * 4 + 4096
    ws = torch.empty(fixed + reduced + int(2.6 * 20_000) * NT * 16, dtype=torch.uint8, device="cuda")
    assert ws.numel() < engine.hist_s3_ws_bytes(R, N, S)                   # (less than one chunk for the whole call)
    try:
        for force in (2, 1, 0):
            engine._abi.call("epg_test_force", 2, force)
            got = engine.hist_s3(X, N, S, ws=ws)
            assert torch.equal(got, want), (force, dirty)
            got2 = engine.hist_s3(X, N, S, counts=got, ws=ws)              # back to back on the same buffers, accumulating
            assert torch.equal(got2, 2 * want), (force, dirty)
    finally:
        engine._abi.call("epg_test_force", 2, 0)


@pytest.mark.parametrize("N,R,pitch_extra,off", [(33, 130, 0, 0), (33, 130, 0, 5), (70, 257, 3, 1), (129, 64, 0, 0), (17, 1000, 15, 7), (64, 65, 0, 3)])
def test_s3_on_packed_and_misaligned_matrices(N, R, pitch_extra, off):
    """The S3 passes start from a transposed copy of the state matrix (k_transpose_states16: 16-byte loads of a bin's states): row
    pitches that are not a multiple of 16 (a packed matrix: pitch = N), bases that are not 16-byte aligned and rows whose last
    16-byte piece would reach past the pitch must give the counts and scores of the padded layout."""
    from epilogos_amd import engine
    engine.require_gpu()
    S = 18
    rng = np.random.default_rng(N * 1000 + R)
    x = rng.choice(S, size=(R, N), p=rng.dirichlet(np.full(S, 0.5))).astype(np.int8)
    pitch = N + pitch_extra
    flat = torch.full((R * pitch + off + 64,), -1, dtype=torch.int8, device="cuda")
    view = flat[off:off + R * pitch].view(R, pitch)
    view[:, :N] = torch.from_numpy(x).cuda()
    Xpad = engine.states_to_device(x)
    c_ref = engine.hist_s3(Xpad, N, S)
    c = engine.hist_s3(view, N, S)
    assert torch.equal(c, c_ref)
    if N * N * S * S * R < 4e8:
        assert np.array_equal(c.cpu().numpy().reshape(N, N, S, S), onp.expected_s3(x, S))
    q = engine.normalise(c)
    a32, a64 = engine.score_s3(view, N, S, q, want32=True, want64=True)
    b32, b64 = engine.score_s3(Xpad, N, S, q, want32=True, want64=True)
    assert torch.equal(a64, b64) and torch.equal(a32, b32)

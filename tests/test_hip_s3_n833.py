"""GPU parity of the S3 path at the BASELINE.json shape: 833 biosamples x 18 states (config 4).

At N = 833 the matrix-core expected kernel runs 157 x 157 block pairs of 96 (sample, state) rows with a ragged last block
(14 994 = 156 * 96 + 18), a K split combined by int32 atomics and the eight-wave shared-A path; the LDS-counter kernel runs
three bin slices (R > 2 * 65 535); the score kernel streams 209 phases of four biosamples with Nceil = 836 zero rows and its
blocks (one per biosample) meet in the same output cells.  None of that is reached by the N <= 129 cases of
test_hip_s3_null.py.  Reference: expected.py:183-200 (s3Calc), scores.py:474-504 (s3Score).

The oracle is the CPU restatement (oracle/oracle_np.py): the full 14 994 x 14 994 co-occurrence matrix through a float32
syrk (exact integers), scores of scattered bins through the closed-form float64 sum.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle_np as onp
from tests.conftest import synth_states

pytestmark = pytest.mark.gpu
S, N = 18, 833
R = 2 * 65536 + 37                      # three slices of the LDS kernel, not a multiple of any tile / K step
BAD = ((70001, 416, -1), (R - 1, 832, 25), (5, 0, 31))   # bytes that are not states: skipped in every pair they are part of


@pytest.fixture(scope="module")
def eng():
    from epilogos_amd import engine
    engine.require_gpu()
    return engine


@pytest.fixture(scope="module")
def case(eng):
    x = synth_states(R, N, seed=833)
    for r, c, v in BAD:
        x[r, c] = v
    X = eng.states_to_device(x)
    c_mfma = eng.hist_s3(X, N, S)                          # matrix-core path (workspace for the transposed matrix)
    c_lds = eng.hist_s3(X, N, S, use_workspace=False)      # LDS-counter path
    torch.cuda.synchronize()
    return {"x": x, "X": X, "mfma": c_mfma, "lds": c_lds}


@pytest.fixture(scope="module")
def oracle_counts(case):
    return onp.expected_s3(case["x"], S)                   # int32 [833, 833, 18, 18]


def test_s3_counts_two_kernels_agree(case):
    assert case["mfma"].dtype == torch.int32
    assert torch.equal(case["mfma"], case["lds"])


def test_s3_counts_invariants(case):
    c = case["mfma"].view(N, N, S, S)
    valid = torch.from_numpy(((case["x"] >= 0) & (case["x"] < S))).cuda()
    nvalid = valid.sum(dim=1).to(torch.int64)              # valid states per bin
    assert int(c.sum(dtype=torch.int64)) == int((nvalid * (nvalid - 1)).sum())     # ordered pairs of valid states
    assert int(torch.diagonal(c, dim1=0, dim2=1).abs().sum()) == 0                 # a == b stays zero (expected.py:185)
    assert torch.equal(c, c.permute(1, 0, 3, 2))                                   # C[a,b,i,j] == C[b,a,j,i]


def test_s3_counts_oracle_n833(case, oracle_counts):
    got = case["mfma"].cpu().numpy().reshape(N, N, S, S)
    assert np.array_equal(got, oracle_counts)


def test_s3_reduced_contraction(eng, case, oracle_counts, monkeypatch):
    """The contraction without each biosample's state S - 1 (its cells follow from the marginals, epg_s3_gemm.hip) against the
    full one: on the matrix WITH bytes that are not states the device-side flag must send the call down the full path, on the
    cleaned matrix the reduced path runs -- the same integers as the oracle either way, and accumulating into non-zero counts."""
    eng._abi.call("epg_test_force", 2, 2)                        # the reduced contraction whatever the call's size
    dirty = eng.hist_s3(case["X"], N, S)
    assert torch.equal(dirty, case["mfma"])
    x = case["x"].copy()
    fix = {}
    for r, c, _v in BAD:
        fix[(r, c)] = (r * 7 + c) % S
        x[r, c] = fix[(r, c)]
    Xc = eng.states_to_device(x)
    red = eng.hist_s3(Xc, N, S)                                  # reduced contraction + reconstruction
    red2 = eng.hist_s3(Xc, N, S, counts=case["mfma"].clone())    # += into non-zero counts
    eng._abi.call("epg_test_force", 2, 1)                        # the contraction over all S states
    full = eng.hist_s3(Xc, N, S)
    eng._abi.call("epg_test_force", 2, 0)
    assert torch.equal(red, full)
    assert torch.equal(red2, full + case["mfma"])
    assert int(red.sum(dtype=torch.int64)) == R * N * (N - 1)
    # against the oracle: the cleaned matrix differs from the fixture's in three bytes, i.e. in the pairs of three bins
    want = oracle_counts.astype(np.int64).copy()
    for (r, c), v in fix.items():
        row = x[r]
        for b in range(N):
            if b != c:
                want[c, b, v, row[b]] += 1
                want[b, c, row[b], v] += 1
    # (5, 0) and (R - 1, 832), (70001, 416) are in different bins: no pair involves two fixed bytes
    assert np.array_equal(red.cpu().numpy().reshape(N, N, S, S), want)


def test_s3_counts_accumulate(eng, case, oracle_counts):
    c = case["mfma"].clone()
    eng.hist_s3(case["X"], N, S, counts=c)                 # += like expectedCombination.py:30-35
    assert torch.equal(c, 2 * case["mfma"])


def test_s3_exp_freq_n833(eng, case, oracle_counts):
    q = eng.normalise(case["mfma"]).cpu().numpy().reshape(N, N, S, S)
    assert np.array_equal(q, onp.normalise(oracle_counts))  # bit-exact float32 (expectedCombination.py:42)


def test_s3_scores_n833(eng, case, oracle_counts):
    x = case["x"]
    qd = eng.normalise(case["mfma"])
    o32, o64 = eng.score_s3(case["X"], N, S, qd, want32=True, want64=True)
    o32b, o64b = eng.score_s3(case["X"], N, S, qd, want32=True, want64=True)
    o32c, _ = eng.score_s3(case["X"], N, S, qd, want32=True, want64=False)
    torch.cuda.synchronize()
    # fixed-point accumulation: run-to-run bit-identical (like S1 / S2), float32 the rounding of the float64
    assert torch.equal(o64, o64b) and torch.equal(o32, o32b) and torch.equal(o32, o32c)
    assert torch.equal(o32, o64.to(torch.float32))
    assert bool(torch.isfinite(o64).all())
    # 96 scattered bins (first, last, both sides of every 4096-bin block border of the score kernel's slices) against the
    # float64 closed form; rows holding a byte that is not a state are left out (the reference's gather is undefined there)
    rng = np.random.default_rng(7)
    rows = set(rng.integers(0, R, size=80).tolist()) | {0, 1, 4095, 4096, 8191, 8192, 65535, 65536, 131071, 131072, R - 2}
    rows -= {r for r, _, _ in BAD}
    rows = np.array(sorted(rows))
    q = qd.cpu().numpy().reshape(N, N, S, S)
    ref = onp.score_s3_f64(x[rows], q, S)
    np.testing.assert_allclose(o64.cpu().numpy()[rows], ref, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(o32.cpu().numpy()[rows], ref.astype(np.float32), rtol=1e-6, atol=1e-9)
    # a byte that is not a state adds nothing: the bin's total is that of its valid biosamples (checked against the oracle
    # on the matrix with that biosample's column dropped for the bin: same as removing its terms)
    r, cidx, _ = BAD[0]
    xr = x[r].copy()
    a_idx, b_idx = np.nonzero(~np.eye(N, dtype=bool))
    keep = (a_idx != cidx) & (b_idx != cidx)
    T = onp.s3_table(q, N).astype(np.float64)
    want = np.zeros(S)
    np.add.at(want, xr[b_idx[keep]], T[a_idx[keep], b_idx[keep], xr[a_idx[keep]], xr[b_idx[keep]]])
    np.testing.assert_allclose(o64.cpu().numpy()[r], want, rtol=1e-6, atol=1e-9)


def test_s3_score_two_kernels_agree(eng, case, oracle_counts, monkeypatch):
    """The biosample-lane kernel (default, 32-bit fixed-point table, epg_s3_lanes.hip) against the bin-lane kernel (float32
    table, float64 folds, epg_test_force(1, 1)) on the same 4400 bins -- three workgroup slices of 1440 with a ragged tail,
    27 chunks of 32 biosamples with a ragged last one (833 = 26 * 32 + 1), bins holding bytes that are not states."""
    q = torch.from_numpy(onp.normalise(oracle_counts).reshape(-1)).cuda()
    lo = 70001 - 2000
    X = case["X"][lo:lo + 4400]
    a32, a64 = eng.score_s3(X, N, S, q, want32=True, want64=True)
    a32b, _ = eng.score_s3(X, N, S, q, want32=True, want64=False)
    assert torch.equal(a32, a32b)                                   # integer accumulation: run-to-run identical
    eng._abi.call("epg_test_force", 1, 1)
    try:
        b32, b64 = eng.score_s3(X, N, S, q, want32=True, want64=True)
    finally:
        eng._abi.call("epg_test_force", 1, 0)
    np.testing.assert_allclose(a64.cpu().numpy(), b64.cpu().numpy(), rtol=1e-6, atol=1e-9)
    assert not torch.equal(a64, b64)                                # two different kernels did run
    # a partition of the bins gives the same bits (the cells are integers until the last step)
    p32, _ = eng.score_s3(X[:1500], N, S, q, want32=True, want64=False)
    assert torch.equal(p32, a32[:1500])

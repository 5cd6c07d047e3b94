"""GPU test at a size where every flat index of the S1 / S2 path passes 2^31: R = 2^27 + 5 bins x 127 biosamples (the Roadmap
width) x 18 states -- 17 GB of states, R * S = 2.4e9 histogram cells, R * S * 4 = 9.7e9 bytes of scores.  Size-independent
properties (the oracle would need days), torch reductions as the independent checker, slices against separate calls."""
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
S, N, R = 18, 127, (1 << 27) + 5


@pytest.fixture(scope="module")
def big():
    import bench
    from epilogos_amd import engine
    engine.require_gpu()
    free, _ = torch.cuda.mem_get_info()
    if free < (60 << 30):
        pytest.skip("needs 60 GB of device memory")
    X = engine.alloc_states(R, N)
    bench.generate_shard(torch, X, N, S, 0)
    H, counts = engine.bin_hist(X, N, S)
    torch.cuda.synchronize()
    yield engine, X, H, counts
    del X, H
    torch.cuda.empty_cache()


def _chunks(n, step=1 << 24):
    for lo in range(0, n, step):
        yield lo, min(lo + step, n)


def test_counts_and_histograms_past_2_31(big):
    eng, X, H, counts = big
    assert R * S > (1 << 31) and X.numel() > (1 << 32)
    assert int(counts.sum().item()) == R * N
    col = torch.zeros(S, dtype=torch.int64, device="cuda")
    for lo, hi in _chunks(R):
        h = H[lo:hi].to(torch.int32)
        rows = h.sum(dim=1)
        assert int(rows.min().item()) == N and int(rows.max().item()) == N      # every bin's histogram sums to N
        col += h.sum(dim=0, dtype=torch.int64)
    assert torch.equal(col, counts)
    # the last rows (flat offsets > 2^31 in H, > 2^34 in X) against a call on the slice alone
    tail = slice(R - 70001, R)
    Ht, ct = eng.bin_hist(X[tail], N, S)
    assert torch.equal(Ht, H[tail])
    assert int(ct.sum().item()) == 70001 * N


def test_s1_scores_past_2_31(big):
    eng, X, H, counts = big
    q = eng.normalise(counts)
    o32, _ = eng.score_s1_from_binhist(H, N, S, q)
    assert o32.shape == (R, S)
    for sl in (slice(0, 5000), slice((1 << 31) // S - 2500, (1 << 31) // S + 2500), slice(R - 5000, R)):
        f32, _ = eng.score_s1(X[sl], N, S, q)                       # fused route on the slice alone
        assert torch.equal(o32[sl], f32)
    assert bool(torch.isfinite(o32[:: 4097]).all())
    del o32


def test_s2_counts_and_scores_past_2_31(big):
    eng, X, H, counts = big
    c2 = eng.hist_s2_from_binhist(H, S)
    assert int(c2.sum().item()) == R * N * (N - 1)                   # ordered pairs of biosamples per bin
    c2 = c2.view(S, S)
    assert torch.equal(c2, c2.t())
    part = torch.zeros(S * S, dtype=torch.int64, device="cuda")
    for lo, hi in _chunks(R, 1 << 25):
        eng.hist_s2_from_binhist(H[lo:hi], S, counts=part)           # additive over bin ranges (the multi-GPU partition)
    assert torch.equal(part.view(S, S), c2)
    q2 = eng.normalise(c2.reshape(-1))
    sl = slice(R - 4096, R)
    a32, _ = eng.score_s2_from_binhist(H[sl], N, S, q2)
    lo = (1 << 31) // S - 5000
    big32, _ = eng.score_s2_from_binhist(H[lo:], N, S, q2)           # a call whose output offsets pass 2^31 elements
    assert torch.equal(big32[-4096:], a32)

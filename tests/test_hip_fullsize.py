"""GPU tests at BASELINE.json's full size (15 M bins x 833 biosamples x 18 states, synthetic, generated on device):
size-independent properties instead of an oracle comparison (the oracle would need hours).  torch reductions are the
independent checker."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
S, N, R = 18, 833, 15_000_000


@pytest.fixture(scope="module")
def world():
    import bench
    from epilogos_amd import engine
    engine.require_gpu()
    X = engine.alloc_states(R, N)
    bench.generate_shard(torch, X, N, S, 0)
    H, counts = engine.bin_hist(X, N, S)
    torch.cuda.synchronize()
    return engine, X, H, counts


def test_every_state_byte_is_counted_once(world):
    eng, X, H, counts = world
    assert int(counts.sum().item()) == R * N                         # nothing lost, nothing double counted
    rows = H.to(torch.int32).sum(dim=1)
    assert int(rows.min().item()) == N and int(rows.max().item()) == N    # every bin's histogram sums to N
    colsum = H.to(torch.int64).sum(dim=0)                            # independent reduction of the cached histograms
    assert torch.equal(colsum, counts)
    # checksum of checksums against torch's own count of each state in the padded matrix
    for s in (0, 5, 17):
        assert int((X[:, :N] == s).sum().item()) == int(counts[s].item())


def test_counts_are_additive_over_bin_ranges(world):
    """The multi-GPU partition rule: counts of contiguous ranges add up to the whole (helpers.splitRows, 8 ranks)."""
    eng, X, H, counts = world
    from epilogos_amd.helpers import splitRows
    acc = torch.zeros(S, dtype=torch.int64, device="cuda")
    for lo, hi in splitRows(R, 8):
        Hs, _ = eng.bin_hist(X[lo:hi], N, S, counts=acc)
        if lo in (0, splitRows(R, 8)[5][0]):
            assert torch.equal(Hs, H[lo:hi])                          # a shard's histograms are the slice of the whole
    assert torch.equal(acc, counts)


def test_s1_scores_full_size(world):
    eng, X, H, counts = world
    q = eng.normalise(counts)
    assert abs(float(q.sum().item()) - 1.0) < 1e-6
    a32, _ = eng.score_s1_from_binhist(H, N, S, q)
    b32, _ = eng.score_s1(X, N, S, q)
    assert torch.equal(a32, b32)                                     # cached-histogram and fused routes: identical bits
    assert bool(torch.isfinite(a32).all())
    # KL property: sum_s p_s log2(p_s/q_s) >= 0 for every bin (float32 rounding of each term allowed)
    assert float(a32.sum(dim=1).min().item()) > -1e-5
    # spot check 4096 scattered bins against the float64 definition evaluated with torch
    idx = torch.arange(0, R, R // 4096, device="cuda")[:4096]
    p = H[idx].to(torch.float64) / N
    qq = q.to(torch.float64)
    ref = torch.where(p > 0, p * torch.log2(p / qq), torch.zeros_like(p))
    torch.testing.assert_close(a32[idx].to(torch.float64), ref, rtol=1e-6, atol=1e-7)


def test_s2_counts_full_size(world):
    eng, X, H, counts = world
    c2 = eng.hist_s2_from_binhist(H, S)
    assert int(c2.sum().item()) == R * N * (N - 1)                   # ordered pairs of distinct biosamples per bin
    c2m = c2.view(S, S)
    assert torch.equal(c2m, c2m.t())                                 # symmetric
    # row sums: sum_j C[i,j] = (N-1) * sum_b h_i
    assert torch.equal(c2m.sum(dim=1), counts * (N - 1))
    q2 = eng.normalise(c2)
    sub = H[:200_000]
    o32, o64 = eng.score_s2_from_binhist(sub, N, S, q2, want32=True, want64=True)
    assert bool(torch.isfinite(o64).all())
    assert float(o64.sum(dim=1).min().item()) > -1e-9                # a KL divergence per bin


def test_placed_histogram_cache(world):
    """engine.place_hist: the histogram cache in another memory class than the matrix -- a view of a held block, the same
    integers as a plain allocation, a report of what was tried, and never slower than the first candidate."""
    eng, X, H, counts = world
    Hp, rep = eng.place_hist(X, N, S)
    assert Hp.shape == (R, S) and Hp.dtype == torch.int16 and Hp.is_contiguous()
    acc = torch.zeros(S, dtype=torch.int64, device="cuda")
    eng.bin_hist(X, N, S, counts=acc, H=Hp)
    assert torch.equal(Hp, H) and torch.equal(acc, counts)
    assert 1 <= rep["blocks_tried"] <= 40 and 0 <= rep["picked"] < rep["blocks_tried"]
    assert rep["ms_picked"] <= rep[[k for k in rep if k.startswith("ms_with_H")][0]][0] + 1e-9
    small, rep1 = eng.place_hist(X[:1000], N, S)                      # under 1 GiB: a plain allocation
    assert rep1 == {"tries": 1} and small.shape == (1000, S)
    # park=True: what the search did not keep stays allocated until release_parked() (bench.py: freed memory is scrubbed in
    # the background at the expense of whatever runs next)
    free0, _ = torch.cuda.mem_get_info()
    Hq, rep2 = eng.place_hist(X, N, S, park=True)
    assert rep2["parked_GiB"] == round((rep2["blocks_tried"] - 1) * rep2["block_GiB"], 1)
    eng.bin_hist(X, N, S, counts=torch.zeros(S, dtype=torch.int64, device="cuda"), H=Hq)
    assert torch.equal(Hq, H)
    del Hq
    eng.release_parked()
    assert not eng._parked
    free1, _ = torch.cuda.mem_get_info()
    assert free1 >= free0 - (1 << 30)                                 # everything went back


def test_paired_job_full_size_properties(world):
    """BASELINE config 5 at full size (15 M bins x (379 + 342) columns of the same synthetic matrix) through the command line's
    session, fed in 24 chromosome-sized parts like bench.py does: size-independent properties.  Every null draw is a
    permutation's bookkeeping (per bin: the two null groups add up to the two real ones and have the groups' sizes); a group
    compared with ITSELF has zero deltas, zero distances and null groups that are still a valid split; the whole job from
    parts equals the job from one part bit for bit (the shuffle key is (file, row), not the launch geometry); the quiescence
    mask is exactly "both groups all-quiescent"."""
    import bench
    from epilogos_amd import backend
    from epilogos_amd.driver import shuffle_key
    eng, X, H, counts = world
    NA, NB = 379, 342
    be = backend.HipBackend()
    XA, XB = X[:, :NA], X[:, 384:384 + NB]                          # column windows of the resident matrix (16-byte aligned starts)
    parts = bench.chromosome_parts(R, 0, R)

    def run(groups, part_list):
        sess = be.open_paired(S, 1, S - 1, -1, 424242)
        pids = [sess.add_staged(groups[0][a:b], NA, groups[1][a:b], NB if groups[1] is XB else NA, shuffle_key(f, r0))
                for f, r0, a, b in part_list]
        sess.ensure_acc(NA + NB)
        sess.finish_device(sum(b - a for _f, _r0, a, b in part_list), NA + (NB if groups[1] is XB else NA))
        nulls = [sess._null_of(p) for p in pids]
        hists = [(sess.parts[p][2], sess.parts[p][3]) for p in pids]
        res = sess.results_device_all(pids)
        sess.check()
        return res, nulls, hists

    res, nulls, hists = run((XA, XB), parts)
    assert sum(r["delta"].shape[0] for r in res) == R
    for (HnA, HnB), (HA, HB) in zip(nulls, hists):
        assert torch.equal(HnA.to(torch.int32) + HnB.to(torch.int32), HA.to(torch.int32) + HB.to(torch.int32))
        assert int(HnA.to(torch.int32).sum(dim=1).min()) == NA == int(HnA.to(torch.int32).sum(dim=1).max())
        assert int(HnB.to(torch.int32).sum(dim=1).min()) == NB == int(HnB.to(torch.int32).sum(dim=1).max())
    delta = torch.cat([r["delta"] for r in res])
    null = torch.cat([r["null"] for r in res])
    quies = torch.cat([r["quies"] for r in res])
    assert bool(torch.isfinite(delta).all()) and bool(torch.isfinite(null).all())
    want_q = (torch.cat([h[0] for h in hists])[:, S - 1] == NA) & (torch.cat([h[1] for h in hists])[:, S - 1] == NB)
    assert torch.equal(quies.bool(), want_q)
    # the sign of a null distance is the sign of KL(null A) - KL(null B): the SMALLER group's plug-in KL is biased upwards (by about
    # (S - 1) / (2 n ln 2)), so with 379 against 342 columns somewhat fewer than half are positive; with equal sizes it is a coin
    frac_pos = float((null > 0).double().mean())
    assert 0.30 < frac_pos < 0.50, frac_pos
    # one part instead of 24: the same job -- per-bin results do not depend on how the bins were cut into launches ... as long
    # as the keys are the same: feed the one part's rows with their (file, row) keys by running it as the 24 slices of ONE call
    # order reversed (launch order must not matter either)
    res_rev, _n, _h = run((XA, XB), parts[::-1])
    assert torch.equal(torch.cat([r["delta"] for r in res_rev[::-1]]), delta)
    assert torch.equal(torch.cat([r["null"] for r in res_rev[::-1]]), null)
    del res_rev, res, nulls, hists, delta, null
    torch.cuda.empty_cache()
    # a group against itself
    res2, nulls2, hists2 = run((XA, XA), parts[:6])
    for r in res2:
        assert int((r["delta"] != 0).sum()) == 0 and int((r["rdist"] != 0).sum()) == 0
    null2 = torch.cat([r["null"] for r in res2])
    frac2 = float((null2 > 0).double().mean() / (null2 != 0).double().mean())
    assert 0.48 < frac2 < 0.52, frac2                               # equal group sizes: the sign is a fair coin
    for (HnA, HnB), (HA, HB) in zip(nulls2, hists2):
        assert torch.equal(HnA.to(torch.int32) + HnB.to(torch.int32), 2 * HA.to(torch.int32))

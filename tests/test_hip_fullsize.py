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
    # 4096 scattered bins (their state rows downloaded) through the oracle: scores.py:317,327-344 restated in numpy
    from oracle import oracle_np as onp
    idx = torch.arange(0, R, R // 4096, device="cuda")[:4096]
    xs = X[idx][:, :N].cpu().numpy()
    _, a64 = eng.score_s1_from_binhist(H[idx].contiguous(), N, S, q, want32=False, want64=True)
    ref = onp.score_s1(xs, q.cpu().numpy(), S)
    np.testing.assert_allclose(a64.cpu().numpy(), ref, rtol=1e-6, atol=0)            # north_star: 1e-6 relative on float64
    np.testing.assert_allclose(a32[idx].cpu().numpy(), ref.astype(np.float32), rtol=2e-7, atol=0)


def test_s2_counts_full_size(world):
    eng, X, H, counts = world
    c2 = eng.hist_s2_from_binhist(H, S)
    assert int(c2.sum().item()) == R * N * (N - 1)                   # ordered pairs of distinct biosamples per bin
    c2m = c2.view(S, S)
    assert torch.equal(c2m, c2m.t())                                 # symmetric
    # row sums: sum_j C[i,j] = (N-1) * sum_b h_i
    assert torch.equal(c2m.sum(dim=1), counts * (N - 1))
    q2 = eng.normalise(c2)
    o32, _ = eng.score_s2_from_binhist(H, N, S, q2)                   # all 15 M bins
    assert bool(torch.isfinite(o32).all())
    assert float(o32.sum(dim=1).min().item()) > -1e-4                # a KL divergence per bin (float32 store of each state's sum)
    # 4096 scattered bins of the 15 M-bin job against the oracle's numbers (scores.py:404-412,426-452 restated in numpy)
    from oracle import oracle_np as onp
    idx = torch.arange(0, R, R // 4096, device="cuda")[:4096]
    xs = X[idx][:, :N].cpu().numpy()
    ref = onp.score_s2(xs, q2.cpu().numpy().reshape(S, S), S)
    _, o64 = eng.score_s2_from_binhist(H[idx].contiguous(), N, S, q2, want32=False, want64=True)
    np.testing.assert_allclose(o64.cpu().numpy(), ref, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(o32[idx].cpu().numpy(), ref.astype(np.float32), rtol=1e-6, atol=1e-9)


def test_placed_histogram_cache(world, monkeypatch):
    """engine.alloc_hist: the histogram cache of a resident matrix in another memory class than the matrix -- a view of the
    head of the device's home block, the same integers as a plain allocation, a report of what was tried (a bounded walk, a
    relative decision, one confirmation over the whole matrix); the same matrix gets the home again without a probe -- also on
    another stream, ordered behind the previous holder's --, a second request while the first is alive gets a plain allocation,
    a matrix under 1 GiB never searches; a session's add_device uses it."""
    from epilogos_amd import backend
    eng, X, H, counts = world
    eng.release_placement()
    first = eng.alloc_hist(X, N, S)                                  # job 1 on a matrix: a plain allocation, no search
    assert eng.placement_report() == {"jobs_seen": 1, "tier": "none yet"} and first.shape == (R, S)
    del first
    eng.release_placement()
    monkeypatch.setenv("EPILOGOS_PLACEMENT_EAGER", "1")              # here: both searches at once (the policy itself: tests/test_host_logic.py)
    Hp = eng.alloc_hist(X, N, S)
    rep = eng.placement_report()
    assert rep["tier"] in ("quick", "deep") and (rep["tier"] == "quick" or not rep["quick"]["blocks_tried"] < 0)
    assert Hp.shape == (R, S) and Hp.dtype == torch.int16 and Hp.is_contiguous() and Hp.data_ptr() % 16 == 0
    acc = torch.zeros(S, dtype=torch.int64, device="cuda")
    eng.bin_hist(X, N, S, counts=acc, H=Hp)
    assert torch.equal(Hp, H) and torch.equal(acc, counts)
    # the walk is bounded, the decision says what it was, and the whole-matrix comparison with the plain allocation settled it
    assert 0 <= rep["blocks_tried"] <= eng.PLACE_DEEP_TRIES and 0 <= rep["picked"] <= rep["blocks_tried"]
    assert rep["probe_device_ms"] <= 6 * eng.PLACE_BUDGET_MS + 15 and rep["walked_GiB"] <= 4.0 * eng.PLACE_DEEP_TRIES
    assert rep["search_ms"] <= eng.PLACE_DEEP_WALL_MS + 1500
    assert all(r > 1.0 for r in rep["ratios"]) and len(rep["ratios"]) == rep["blocks_tried"] + 1
    assert rep["good"] == (rep["picked"] != 0)
    if rep["good"]:
        wm = rep["whole_matrix_ms"]
        assert wm[str(rep["picked"])] < wm["0"] * (1 - eng.PLACE_WIN) and rep["picked"] not in rep["lost_over_the_whole_matrix"]
    home = Hp.data_ptr()
    if not rep["good"]:                                              # this box kept the plain allocation: nothing more to hand out
        del Hp
        again = eng.alloc_hist(X, N, S)
        assert eng.placement_report()["jobs_seen"] >= 2 and again.shape == (R, S)
        del again
        eng.release_placement()
        torch.cuda.empty_cache()
        return
    other = eng.alloc_hist(X, N, S)                                   # the home is in use: a plain allocation
    assert other.data_ptr() != home and eng.placement_report()["plain_while_home_in_use"] == 1
    del Hp, other
    again = eng.alloc_hist(X[: R // 2], N, S)                          # a view of the same matrix: the home, no probe
    assert again.data_ptr() == home and again.shape == (R // 2, S) and eng.placement_report()["reuses"] == 1
    del again
    side = torch.cuda.Stream()                                        # another stream takes the home: it waits for the previous holder's stream
    eng.bin_hist(X, N, S, counts=torch.zeros(S, dtype=torch.int64, device="cuda"), H=eng.alloc_hist(X, N, S))   # (main stream: still running ...)
    with torch.cuda.stream(side):
        Hs2 = eng.alloc_hist(X, N, S)                                 # ... when the side stream gets the same block and overwrites it
        assert Hs2.data_ptr() == home
        acc2 = torch.zeros(S, dtype=torch.int64, device="cuda")
        eng.bin_hist(X, N, S, counts=acc2, H=Hs2)
    side.synchronize()
    assert torch.equal(Hs2, H) and torch.equal(acc2, counts)
    del Hs2
    small = eng.alloc_hist(X[:1000].clone(), N, S)                    # under 1 GiB: plain, the home stays
    assert small.shape == (1000, S) and small.data_ptr() != home
    be = backend.HipBackend()
    sess = be.open_single(S, 1)
    pid = sess.add_device(X, N)
    assert sess.parts[pid].data_ptr() == home and torch.equal(sess.parts[pid], H)
    sess2 = be.open_single(S, 2)                                      # (S2 sessions place too; this one finds the home taken)
    pid2 = sess2.add_device(X, N)
    assert sess2.parts[pid2].data_ptr() != home and torch.equal(sess2.parts[pid2], H)
    del sess, sess2
    Y = X[: R // 4].clone()                                           # another matrix: one probe of the home, kept or replaced
    Hy = eng.alloc_hist(Y, N, S)
    eng.bin_hist(Y, N, S, counts=torch.zeros(S, dtype=torch.int64, device="cuda"), H=Hy)
    assert torch.equal(Hy, H[: R // 4])
    r2 = eng.placement_report()
    assert r2.get("revalidated", 0) == 1 or r2["blocks_tried"] >= 0
    del Hy, Y
    eng.release_placement()
    assert eng.placement_report() is None
    import os                                                         # no block to be had (here: none allowed): a plain allocation
    os.environ["EPILOGOS_PLACEMENT_TRIES"] = "0"
    try:
        Hn = eng.alloc_hist(X, N, S)
    finally:
        del os.environ["EPILOGOS_PLACEMENT_TRIES"]
    rn = eng.placement_report()
    assert Hn.shape == (R, S) and rn["blocks_tried"] == 0 and not rn["good"] and rn["picked"] == 0
    eng.release_placement()
    torch.cuda.empty_cache()                                          # (the walked blocks back to the driver: the tests behind this one want the memory)


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


def test_s3_job_full_size(world):
    """BASELINE config 4 at full size (15 M bins x 833 x 18) through the command line's session: the expected counts by their
    invariants (expected.py:183-200: every ordered pair of distinct biosamples once per bin, a zero diagonal, C[a,b,i,j] ==
    C[b,a,j,i], marginals that are the state counts of a column), the scores of the whole job (scores.py:474-504) finite and
    bit-identical to the scores of the two halves of the genome scored on their own (the cells are integers until the last
    step: no dependence on the partition -- what the 8-GPU split relies on), and 32 scattered bins against the float64 oracle
    at the contract's 1e-6."""
    from oracle import oracle_np as onp
    from epilogos_amd import backend
    eng, X, H, counts = world
    del H
    torch.cuda.empty_cache()
    be = backend.HipBackend()
    sess = be.open_single(S, 3)
    pid = sess.add_device(X, N)                                      # STEP 1: the fp4 one-hot contraction, seven chunks of 2 M bins + one
    c = sess.acc.view(N, N, S, S)
    assert int(sess.acc.sum(dtype=torch.int64)) == R * N * (N - 1)
    assert int(torch.diagonal(c, dim1=0, dim2=1).abs().sum()) == 0
    for a, b in ((0, 1), (416, 832), (832, 3)):
        assert torch.equal(c[a, b], c[b, a].t())
        col = X[:, a].to(torch.int64)
        assert torch.equal(c[a, b].sum(dim=1, dtype=torch.int64), torch.bincount(col, minlength=S)[:S])
    assert torch.equal(c.sum(dim=(2, 3), dtype=torch.int64), (torch.ones(N, N, dtype=torch.int64, device="cuda") - torch.eye(N, dtype=torch.int64, device="cuda")) * R)
    sess.finish_device(R, N)                                         # STEP 2 (the count check runs at once for S3)
    q = sess.q
    sess.launch_scores([pid])                                        # STEP 3
    whole = sess.early_scores(pid)
    assert whole.shape == (R, S) and bool(torch.isfinite(whole).all())
    half = R // 2 // 32 * 32 + 5                                     # a cut inside a tile
    for lo, hi in ((0, half), (half, R)):
        p32, _ = eng.score_s3(X[lo:hi], N, S, q, want32=True, want64=False)
        assert torch.equal(p32, whole[lo:hi]), (lo, hi)
        del p32
    idx = torch.arange(0, R, R // 32, device="cuda")[:32]
    xs = X[idx][:, :N].cpu().numpy()
    qh = q.cpu().numpy().reshape(N, N, S, S)
    ref = onp.score_s3_f64(xs, qh, S)
    _, o64 = eng.score_s3(X[idx].contiguous(), N, S, q, want32=False, want64=True)
    np.testing.assert_allclose(o64.cpu().numpy(), ref, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(whole[idx].cpu().numpy(), ref.astype(np.float32), rtol=1e-6, atol=1e-9)
    sess.finish(R, N)

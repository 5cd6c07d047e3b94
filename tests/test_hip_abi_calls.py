"""GPU: every entry point of include/epilogos_amd.h is called at least once straight through ctypes (no engine
helpers), including the error paths a caller can hit."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle_np as onp
from tests.conftest import synth_states

pytestmark = pytest.mark.gpu
S = 18


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def abi():
    from epilogos_amd import _abi, engine
    engine.require_gpu()
    return _abi


def test_direct_calls_match_oracle(abi):
    from epilogos_amd import engine
    R, N = 700, 41
    x = synth_states(R, N, seed=2)
    X = engine.states_to_device(x)
    ldx = X.stride(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert abi.call("epg_version") == abi.ABI_VERSION and abi.call("epg_device_cus") > 0
    # S1 / S2 / S3 expected through the X-taking entry points
    c1 = torch.zeros(S, dtype=torch.int64, device="cuda")
    abi.call("epg_hist_s1", _p(X), R, N, ldx, S, _p(c1), st)
    assert np.array_equal(c1.cpu().numpy(), onp.expected_s1(x, S))
    ws = torch.empty(abi.call("epg_ws_bytes", 2, R, N, S), dtype=torch.uint8, device="cuda")
    c2 = torch.zeros(S * S, dtype=torch.int64, device="cuda")
    abi.call("epg_hist_s2", _p(X), R, N, ldx, S, _p(c2), _p(ws), ws.numel(), st)
    assert np.array_equal(c2.cpu().numpy().reshape(S, S), onp.expected_s2(x, S))
    ws3 = torch.empty(abi.call("epg_ws_bytes", 3, R, N, S), dtype=torch.uint8, device="cuda")
    c3 = torch.zeros(N * N * S * S, dtype=torch.int32, device="cuda")
    abi.call("epg_hist_s3", _p(X), R, N, ldx, S, _p(c3), _p(ws3), ws3.numel(), st)      # matrix-core path
    c3b = torch.zeros_like(c3)
    abi.call("epg_hist_s3", _p(X), R, N, ldx, S, _p(c3b), None, 0, st)                   # LDS-atomic path (no workspace)
    ref3 = onp.expected_s3(x, S)
    assert np.array_equal(c3.cpu().numpy().reshape(ref3.shape), ref3) and torch.equal(c3, c3b)
    # normalise
    q1 = torch.empty(S, dtype=torch.float32, device="cuda")
    wsn = torch.empty(256, dtype=torch.uint8, device="cuda")
    abi.call("epg_normalise_i64", _p(c1), S, _p(q1), _p(wsn), 256, st)
    assert np.array_equal(q1.cpu().numpy(), onp.normalise(onp.expected_s1(x, S)))
    q3 = torch.empty(c3.numel(), dtype=torch.float32, device="cuda")
    abi.call("epg_normalise_i32", _p(c3), c3.numel(), _p(q3), _p(wsn), 256, st)
    assert np.array_equal(q3.cpu().numpy().reshape(ref3.shape), onp.normalise(ref3))
    # workspace too small / bad arguments are reported, not executed
    with pytest.raises(abi.EpilogosHipError) as e:
        abi.call("epg_score_s1", _p(X), R, N, ldx, S, _p(q1), None, _p(torch.empty((R, S), device="cuda")), _p(wsn), 16, st)
    assert e.value.code == -4
    with pytest.raises(abi.EpilogosHipError) as e:
        abi.call("epg_bin_hist", _p(X), R, N, N - 1, S, None, _p(c1), st)
    assert e.value.code == -1
    ws3 = torch.empty(int(abi.call("epg_ws_bytes", 3, R, N, S)) + 64, dtype=torch.uint8, device="cuda")
    for name, args in (("epg_hist_s3", (_p(c3),)), ("epg_score_s3", (_p(q3), None, _p(torch.empty((R, S), device="cuda"))))):
        with pytest.raises(abi.EpilogosHipError) as e:                          # an S3 workspace that is not 16-byte aligned is refused
            abi.call(name, _p(X), R, N, ldx, S, *args, C.c_void_p(ws3.data_ptr() + 8), ws3.numel() - 64, st)
        assert e.value.code == -1 and "aligned" in str(e.value)
    with pytest.raises(abi.EpilogosHipError) as e:
        abi.call("epg_bin_hist", _p(X), R, N, ldx, 128, None, _p(c1), st)      # states are int8: 127 is the largest model
    assert e.value.code == -2
    with pytest.raises(abi.EpilogosHipError) as e:                              # the matrix-scanning null kernel stops at 31 states
        abi.call("epg_null_hist", _p(X), N, ldx, _p(X), N, ldx, R, 40, N, N, 1, 0, _p(torch.empty((R, 40), dtype=torch.int16, device="cuda")),
                 _p(torch.empty((R, 40), dtype=torch.int16, device="cuda")), st)
    assert e.value.code == -2


def test_group_size_option_of_the_null_shuffle(abi):
    """-g/--group-size: the null halves have `g` columns each (helpers.py:190-194); S2 keeps the real groups' P (Q9)."""
    from epilogos_amd import backend
    be = backend.HipBackend()
    xa, xb = synth_states(500, 12, seed=1), synth_states(500, 9, seed=2)
    q1 = onp.normalise(onp.expected_s1(np.concatenate([xa, xb], axis=1), S))
    na, nb = be.null_scores(xa, xb, S, 1, q1, 6, seed=3)
    assert na.shape == (500, S) and np.isfinite(na).all() and np.isfinite(nb).all()
    HA, HB = be.engine.null_hist(be.to_device(xa), 12, be.to_device(xb), 9, S, 6, 6, 3)
    ref = onp.kl(be.engine.hist_to_numpy(HA).astype(np.float64) / 6, q1[None, :]).astype(np.float32)
    np.testing.assert_allclose(na, ref, rtol=2e-7, atol=0)
    q2 = onp.normalise(onp.expected_s2(np.concatenate([xa, xb], axis=1), S))
    na2, _ = be.null_scores(xa, xb, S, 2, q2, 6, seed=3)
    h = be.engine.hist_to_numpy(HA).astype(np.int64)
    num = h[:, :, None] * h[:, None, :]
    idx = np.arange(S)
    num[:, idx, idx] = h * (h - 1)
    ref2 = onp.kl(num / (12 * 11), q2[None]).sum(axis=1)
    np.testing.assert_allclose(na2, ref2.astype(np.float32), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("seed", range(12))
def test_bin_hist_random_geometry(abi, seed):
    """Random R, N, row pitch and base misalignment against the oracle."""
    from epilogos_amd import engine
    rng = np.random.default_rng(seed)
    R, N = int(rng.integers(1, 400)), int(rng.integers(1, 1300))
    pitch = N + int(rng.integers(0, 40))
    off = int(rng.integers(0, 16))
    x = rng.integers(0, S, size=(R, N)).astype(np.int8)
    flat = torch.full((R * pitch + off + 64,), -1, dtype=torch.int8, device="cuda")
    view = flat[off:off + R * pitch].view(R, pitch)
    view[:, :N] = torch.from_numpy(x).cuda()
    H, counts = engine.bin_hist(view, N, S)
    h = onp.bin_hist(x, S)
    assert np.array_equal(engine.hist_to_numpy(H).astype(np.int64), h), (R, N, pitch, off)
    assert np.array_equal(counts.cpu().numpy(), h.sum(axis=0))


def test_combine_score_s1_and_pair_hist_direct(abi):
    """epg_combine_score_s1 (STEP 2 + STEP 3 in one call) equals normalise + score_s1_from_binhist bit for bit and can
    leave the counts zeroed; epg_hist_s2_from_binhist_pair equals the counts of the concatenated matrix."""
    from epilogos_amd import engine
    R, NA, NB = 3001, 37, 22
    xa, xb = synth_states(R, NA, seed=5), synth_states(R, NB, seed=6)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    XA, XB = engine.states_to_device(xa), engine.states_to_device(xb)
    HA, cA = engine.bin_hist(XA, NA, S)
    HB, _ = engine.bin_hist(XB, NB, S)
    # --- combine
    q_ref = engine.normalise(cA)
    o32_ref, o64_ref = engine.score_s1_from_binhist(HA, NA, S, q_ref, want32=True, want64=True)
    counts = cA.clone()
    q = torch.empty(S, dtype=torch.float32, device="cuda")
    o32, o64 = torch.empty_like(o32_ref), torch.empty_like(o64_ref)
    ws = torch.empty(abi.call("epg_ws_bytes", 1, 0, NA, S), dtype=torch.uint8, device="cuda")
    abi.call("epg_combine_score_s1", _p(counts), 0, _p(HA), R, NA, S, _p(q), _p(o64), _p(o32), _p(ws), ws.numel(), st)
    assert torch.equal(q, q_ref) and torch.equal(o32, o32_ref) and torch.equal(o64, o64_ref) and torch.equal(counts, cA)
    assert np.array_equal(q.cpu().numpy(), onp.normalise(onp.expected_s1(xa, S)))
    np.testing.assert_allclose(o64.cpu().numpy(), onp.score_s1(xa, q.cpu().numpy(), S), rtol=1e-11, atol=0)
    abi.call("epg_combine_score_s1", _p(counts), 1, _p(HA), R, NA, S, _p(q), None, _p(o32), _p(ws), ws.numel(), st)
    assert torch.equal(o32, o32_ref) and int(counts.abs().sum()) == 0           # rezero: ready for the next job
    with pytest.raises(abi.EpilogosHipError) as e:
        abi.call("epg_combine_score_s1", _p(counts), 0, _p(HA), R, NA, S, _p(q), None, _p(o32), _p(ws), 16, st)
    assert e.value.code == -4
    # --- the same score pass with a caller-built table (the command line's: numpy on the host): the oracle's bits
    from epilogos_amd.scores import s1ScoreTable
    qh = q_ref.cpu().numpy()
    t64, t32 = s1ScoreTable(qh, NA)
    T64, T32 = torch.from_numpy(t64).cuda(), torch.from_numpy(t32).cuda()
    abi.call("epg_score_s1_from_binhist_table", _p(HA), R, NA, S, _p(T64), _p(T32), _p(o64), _p(o32), st)
    ref = onp.score_s1(xa, qh, S)
    assert np.array_equal(o64.cpu().numpy(), ref) and np.array_equal(o32.cpu().numpy(), ref.astype(np.float32))
    o32.zero_()
    abi.call("epg_score_s1_from_binhist_table", _p(HA), R, NA, S, None, _p(T32), None, _p(o32), st)
    assert np.array_equal(o32.cpu().numpy(), ref.astype(np.float32))
    with pytest.raises(abi.EpilogosHipError):
        abi.call("epg_score_s1_from_binhist_table", _p(HA), R, NA, S, None, _p(T32), _p(o64), _p(o32), st)   # out64 without T64
    # --- S2 counts of [A|B] from the two groups' histograms
    c2 = torch.zeros(S * S, dtype=torch.int64, device="cuda")
    abi.call("epg_hist_s2_from_binhist_pair", _p(HA), _p(HB), R, S, _p(c2), st)
    assert np.array_equal(c2.cpu().numpy().reshape(S, S), onp.expected_s2(np.concatenate([xa, xb], axis=1), S))
    with pytest.raises(abi.EpilogosHipError):
        abi.call("epg_hist_s2_from_binhist_pair", _p(HA), None, R, S, _p(c2), st)
    # --- quiescence from the histograms == from the matrices
    xq, yq = xa.copy(), xb.copy()
    xq[10:20] = 17; yq[10:25] = 17
    XQ, YQ = engine.states_to_device(xq), engine.states_to_device(yq)
    HQ, _ = engine.bin_hist(XQ, NA, S, want_counts=False)
    HY, _ = engine.bin_hist(YQ, NB, S, want_counts=False)
    mq = torch.empty(R, dtype=torch.uint8, device="cuda")
    abi.call("epg_quiescent_from_binhist", _p(HQ), _p(HY), R, S, NA, NB, 17, _p(mq), st)
    assert torch.equal(mq, engine.quiescent(XQ, NA, YQ, NB, 17)) and int(mq.sum()) >= 10
    assert np.array_equal(mq.cpu().numpy().astype(bool), onp.quiescent_mask(xq, yq, 17))
    abi.call("epg_quiescent_from_binhist", _p(HQ), _p(HY), R, S, NA, NB, -1, _p(mq), st)
    assert int(mq.sum()) == 0
    # --- null groups from the real groups' histograms
    OA, OB = torch.empty_like(HA), torch.empty_like(HB)
    abi.call("epg_null_hist_from_binhist", _p(HA), _p(HB), R, S, NA + NB, NA, NB, 77, 0, _p(OA), _p(OB), st)
    tot = engine.hist_to_numpy(HA).astype(np.int64) + engine.hist_to_numpy(HB).astype(np.int64)
    got = engine.hist_to_numpy(OA).astype(np.int64) + engine.hist_to_numpy(OB).astype(np.int64)
    assert np.array_equal(got, tot) and (engine.hist_to_numpy(OA).astype(np.int64).sum(axis=1) == NA).all()
    with pytest.raises(abi.EpilogosHipError) as e:
        abi.call("epg_null_hist_from_binhist", _p(HA), _p(HB), R, S, NA + NB, NA, NB + 1, 77, 0, _p(OA), _p(OB), st)
    assert e.value.code == -1


def test_combine_score_s1_long_table(abi):
    """(N + 1) * S beyond the single-block table kernel: the separate kernels run, same results."""
    from epilogos_amd import engine
    R, N, S_ = 64, 20000, 15
    x = synth_states(R, N, S=S_, seed=9)
    X = engine.states_to_device(x)
    H, c = engine.bin_hist(X, N, S_)
    q_ref = engine.normalise(c)
    o32_ref, _ = engine.score_s1_from_binhist(H, N, S_, q_ref)
    counts = c.clone()
    q, o32, _ = engine.combine_score_s1(counts, H, N, S_, rezero=True)
    assert torch.equal(q, q_ref) and torch.equal(o32, o32_ref) and int(counts.abs().sum()) == 0


@pytest.mark.parametrize("S_", [18, 15, 20, 40])
def test_bin_hist_parts_equals_a_call_per_part(abi, S_):
    """epg_bin_hist_parts straight through ctypes: parts of different row counts (empty, one row, not a multiple of 32), widths
    in several load-schedule classes (1 .. 8 groups per row and the any-width loop), padded and packed row pitches, with and
    without histograms -- the integers of one epg_bin_hist call per part, and of the oracle."""
    from epilogos_amd import engine
    rng = np.random.default_rng(100 + S_)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(0, 40), (1, 40), (95, 379), (64, 342), (33, 379), (200, 120), (7, 1100), (129, 833), (31, 833), (50, 342)]
    xs = [synth_states(r, n, S=S_, seed=int(rng.integers(1 << 30)), uniform=S_ > 18) for r, n in shapes]
    Xs = []
    for k, x in enumerate(xs):
        if k % 3 == 2 and x.shape[0]:                       # a packed matrix (row pitch = width, unaligned rows)
            flat = torch.full((x.size + 64,), -1, dtype=torch.int8, device="cuda")
            v = flat[:x.size].view(x.shape)
            v.copy_(torch.from_numpy(x))
            Xs.append(v)
        else:
            Xs.append(engine.states_to_device(x) if x.shape[0] else torch.empty((0, 48), dtype=torch.int8, device="cuda"))
    n = len(Xs)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() if t is not None and t.numel() else None for t in ts])
    R = (C.c_int64 * n)(*[x.shape[0] for x in xs])
    N = (C.c_int32 * n)(*[x.shape[1] for x in xs])
    ldx = (C.c_int64 * n)(*[X.stride(0) if X.shape[0] else 48 for X in Xs])
    Hs = engine.hist_rows_alloc([x.shape[0] for x in xs], S_, "cuda")
    counts = torch.zeros(S_, dtype=torch.int64, device="cuda")
    abi.call("epg_bin_hist_parts", n, arr(Xs), R, N, ldx, S_, arr(Hs), _p(counts), st)
    want = np.zeros(S_, dtype=np.int64)
    for x, X, H in zip(xs, Xs, Hs):
        if not x.shape[0]:
            continue
        h = onp.bin_hist(x, S_)
        assert np.array_equal(engine.hist_to_numpy(H).astype(np.int64), h), x.shape
        H1, _ = engine.bin_hist(X, x.shape[1], S_, want_counts=False)
        assert torch.equal(H1, H)
        want += h.sum(axis=0)
    assert np.array_equal(counts.cpu().numpy(), want)
    counts.zero_()
    abi.call("epg_bin_hist_parts", n, arr(Xs), R, N, ldx, S_, None, _p(counts), st)          # counts only
    assert np.array_equal(counts.cpu().numpy(), want)
    Hs2, c2 = engine.bin_hist_parts(Xs, [x.shape[1] for x in xs], S_, counts=None)             # the engine wrapper, histograms only
    assert c2 is None and all(torch.equal(a, b) for a, b in zip(Hs, Hs2))
    abi.call("epg_bin_hist_parts", 0, None, None, None, None, S_, None, _p(counts), st)      # no parts: nothing happens
    with pytest.raises(abi.EpilogosHipError) as e:
        bad = (C.c_int64 * n)(*[8] * n)                       # row pitch smaller than the width
        abi.call("epg_bin_hist_parts", n, arr(Xs), R, N, bad, S_, arr(Hs), _p(counts), st)
    assert e.value.code == -1


def test_bin_hist_parts_more_parts_than_one_launch_holds(abi):
    """More parts than the kernel argument holds (48): several launches, same integers."""
    from epilogos_amd import engine
    rng = np.random.default_rng(7)
    xs = [synth_states(int(rng.integers(1, 90)), 61, seed=k) for k in range(110)]
    Xs = [engine.states_to_device(x) for x in xs]
    counts = torch.zeros(S, dtype=torch.int64, device="cuda")
    Hs, _ = engine.bin_hist_parts(Xs, [61] * len(Xs), S, counts=counts)
    allx = np.concatenate(xs)
    assert np.array_equal(counts.cpu().numpy(), onp.expected_s1(allx, S))
    assert np.array_equal(np.concatenate([engine.hist_to_numpy(H) for H in Hs]).astype(np.int64), onp.bin_hist(allx, S))


@pytest.mark.parametrize("ga_gb", [None, (20, 20), (60, 60)])
def test_null_hist_parts_equals_a_call_per_part(abi, ga_gb):
    """epg_null_hist_from_binhist_parts: the null groups of several parts in one launch are, bit for bit, those of a call per
    part with the part's shuffle key (all three sampler kernels: the bit-string one, the two-string one of -g, and -- forced
    through the test hook -- the column-by-column one)."""
    from epilogos_amd import engine
    NA, NB = 70, 53
    ga, gb = ga_gb or (NA, NB)
    rows = [0, 1, 63, 64, 65, 300, 17]
    rng = np.random.default_rng(3)
    HAs, HBs = [], []
    for r in rows:
        xa, xb = synth_states(r, NA, seed=int(rng.integers(1 << 30))), synth_states(r, NB, seed=int(rng.integers(1 << 30)))
        HAs.append(engine.bin_hist(engine.states_to_device(xa), NA, S, want_counts=False)[0] if r else torch.empty((0, S), dtype=torch.int16, device="cuda"))
        HBs.append(engine.bin_hist(engine.states_to_device(xb), NB, S, want_counts=False)[0] if r else torch.empty((0, S), dtype=torch.int16, device="cuda"))
    keys = [(k << 40) + 1000 * k for k in range(len(rows))]
    for force in (0, 1):
        abi.call("epg_test_force", 0, force)
        try:
            OAs, OBs = engine.null_hist_from_binhist_parts(HAs, HBs, NA + NB, S, ga, gb, 4242, keys)
            for HA, HB, OA, OB, key in zip(HAs, HBs, OAs, OBs, keys):
                if not HA.shape[0]:
                    continue
                A1, B1 = engine.null_hist_from_binhist(HA, HB, NA + NB, S, ga, gb, 4242, key)
                assert torch.equal(A1, OA) and torch.equal(B1, OB)
                assert int(OA.to(torch.int32).sum(dim=1).min()) == ga == int(OA.to(torch.int32).sum(dim=1).max())
        finally:
            abi.call("epg_test_force", 0, 0)
    with pytest.raises(abi.EpilogosHipError):
        abi.call("epg_test_force", 99, 1)


@pytest.mark.parametrize("S_,NA,NB", [(18, 379, 342), (18, 70, 53), (15, 130, 200), (25, 300, 257), (18, 512, 400)])
def test_pair_count_null_parts_equals_the_two_kernels(abi, S_, NA, NB):
    """epg_pair_count_null_parts (count pass of both groups + null draw in one kernel) against epg_bin_hist_parts +
    epg_null_hist_from_binhist_parts: the same histograms, the same state counts, the same null groups bit for bit -- parts of
    0, 1, 63, 64, 65 and a few hundred rows, with and without the count accumulator; shapes outside the fused kernel's are
    declined with EPG_ERR_UNSUPPORTED."""
    from epilogos_amd import engine
    rng = np.random.default_rng(S_ * 1000 + NA)
    rows = [0, 1, 63, 64, 65, 300, 129, 1000]
    xs_a = [synth_states(r, NA, S=S_, seed=int(rng.integers(1 << 30)), uniform=S_ > 18) for r in rows]
    xs_b = [synth_states(r, NB, S=S_, seed=int(rng.integers(1 << 30)), uniform=S_ > 18) for r in rows]
    xs_a[5][7, 3] = -1                                          # a column without a state
    xs_b[5][9, :] = 1
    xs_a[5][9, :] = 1                                           # a bin in which every column holds one state: nothing to draw
    XAs = [engine.states_to_device(x) if x.shape[0] else torch.empty((0, engine.padded_width(NA)), dtype=torch.int8, device="cuda") for x in xs_a]
    XBs = [engine.states_to_device(x) if x.shape[0] else torch.empty((0, engine.padded_width(NB)), dtype=torch.int8, device="cuda") for x in xs_b]
    keys = [(k << 40) + 17 * k for k in range(len(rows))]
    c_ref = torch.zeros(S_, dtype=torch.int64, device="cuda")
    H, _ = engine.bin_hist_parts(XAs + XBs, [NA] * len(rows) + [NB] * len(rows), S_, counts=c_ref)
    HA_ref, HB_ref = H[:len(rows)], H[len(rows):]
    OA_ref, OB_ref = engine.null_hist_from_binhist_parts(HA_ref, HB_ref, NA + NB, S_, NA, NB, 4242, keys)
    for with_counts in (True, False):
        c = torch.zeros(S_, dtype=torch.int64, device="cuda") if with_counts else None
        HA, HB, OA, OB = engine.pair_count_null_parts(XAs, XBs, NA, NB, S_, 4242, keys, counts=c)
        for k in range(len(rows)):
            assert torch.equal(HA[k], HA_ref[k]) and torch.equal(HB[k], HB_ref[k]), (k, rows[k])
            assert torch.equal(OA[k], OA_ref[k]) and torch.equal(OB[k], OB_ref[k]), (k, rows[k])
        if with_counts:
            assert torch.equal(c, c_ref)
            want = sum(np.bincount(x[x >= 0].ravel().astype(np.int64), minlength=S_)[:S_] for x in xs_a + xs_b if x.shape[0])
            assert np.array_equal(c.cpu().numpy(), want)
    with pytest.raises(abi.EpilogosHipError) as e:              # widths in different load-schedule classes: the two kernels' job
        engine.pair_count_null_parts([engine.states_to_device(synth_states(5, 100))], [engine.states_to_device(synth_states(5, 300))],
                                     100, 300, 18, 1, [0])
    assert e.value.code == -2
    with pytest.raises(abi.EpilogosHipError) as e:
        engine.pair_count_null_parts([engine.states_to_device(synth_states(5, 40, S=20, uniform=True))],
                                     [engine.states_to_device(synth_states(5, 40, S=20, uniform=True))], 40, 40, 20, 1, [0])
    assert e.value.code == -2


@pytest.mark.parametrize("S_,N,R", [(18, 833, 3000), (18, 41, 1), (18, 379, 33), (15, 127, 2049), (25, 1024, 700), (18, 1030, 300),
                                     (20, 64, 500), (18, 200, 31)])
def test_bin_hist_s2_equals_count_pass_then_pair_count_pass(abi, S_, N, R):
    """epg_bin_hist_s2 (the S2 pair counts folded into the count pass: one launch) against epg_bin_hist + epg_hist_s2_from_binhist and
    the oracle (expected.py:146-158): histograms, pair counts and -- when asked for -- state counts, bit for bit; row counts that
    are not a multiple of 32 (a super-tile's unused rows must not be multiplied), accumulation into non-zero counts, shapes that
    take the two-pass fallback (a 20-state model, 1030 columns), bytes that are not states."""
    from epilogos_amd import engine
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = synth_states(R, N, S=S_, seed=R + N, uniform=S_ > 18)
    if R > 3:
        x[R // 2, N // 2] = -1
    X = engine.states_to_device(x)
    H_ref, c1_ref = engine.bin_hist(X, N, S_)
    c2_ref = engine.hist_s2_from_binhist(H_ref, S_)
    H = torch.empty((R, S_), dtype=torch.int16, device="cuda")
    c1 = torch.zeros(S_, dtype=torch.int64, device="cuda")
    c2 = torch.zeros(S_ * S_, dtype=torch.int64, device="cuda")
    abi.call("epg_bin_hist_s2", _p(X), R, N, X.stride(0), S_, _p(H), _p(c1), _p(c2), st)
    assert torch.equal(H, H_ref) and torch.equal(c1, c1_ref) and torch.equal(c2, c2_ref), (S_, N, R)
    h = engine.hist_to_numpy(H).astype(np.int64)
    want = h.T @ h
    want[np.arange(S_), np.arange(S_)] -= h.sum(axis=0)
    assert np.array_equal(c2.cpu().numpy().reshape(S_, S_), want)
    H2, c2b = engine.bin_hist_s2(X, N, S_, counts2=c2.clone())                      # += into non-zero counts, no state counts
    assert torch.equal(H2, H_ref) and torch.equal(c2b, 2 * c2_ref)
    with pytest.raises(abi.EpilogosHipError):
        abi.call("epg_bin_hist_s2", _p(X), R, N, X.stride(0), S_, None, None, _p(c2), st)      # H is required

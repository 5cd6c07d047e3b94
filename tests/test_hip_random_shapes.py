"""A randomised sweep of shapes through the S1 / S2 / paired entry points against the oracle: state counts with and without a
compile-time instantiation (15, 18, 25 have one; 7, 21, 31 run on the next larger core or the generic kernels), odd and tiny bin
counts (tile tails of 1 .. 63 rows), narrow and wide groups, columns that hold no state.  Round 4 added it after a multi-part
test had found a two-byte tail dropped by a staged loader for odd S x odd rows: shapes nobody had thought of are drawn here."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle_np as onp
from tests.conftest import synth_states

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    import os
    n = int(os.environ.get("EPG_RANDOM_SHAPES", n))              # a longer hunt: EPG_RANDOM_SHAPES=400 pytest ... -m gpu
    seed = seed + int(os.environ.get("EPG_RANDOM_SEED", 0))
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        S = int(rng.choice([7, 15, 18, 21, 25, 31]))
        N = int(rng.choice([2, 3, 5, 16, 17, 33, 64, 100, 127, 128, 129, 255, 379]))
        R = int(rng.choice([1, 2, 63, 64, 65, 127, 191, 317, 500, 1025]))
        out.append((S, N, R, int(rng.integers(1 << 30))))
    return out


@pytest.fixture(scope="module")
def eng():
    from epilogos_amd import engine
    engine.require_gpu()
    return engine


@pytest.mark.parametrize("S,N,R,seed", _cases(28, 4))
def test_single_group_s1_s2_random_shapes(eng, S, N, R, seed):
    x = synth_states(R, N, S=S, seed=seed, uniform=(seed % 3 == 0))
    if R > 2 and N > 2:
        x[R // 2, N // 2] = -1                                   # a column that holds no state in one bin
    X = eng.states_to_device(x)
    H, counts = eng.bin_hist(X, N, S)
    wh = onp.bin_hist(x, S)
    assert np.array_equal(eng.hist_to_numpy(H), wh)
    assert np.array_equal(counts.cpu().numpy(), wh.sum(axis=0))
    xv = np.where(x < 0, 0, x)                                   # scores: the oracle needs states everywhere
    Xv = eng.states_to_device(xv)
    Hv, cv = eng.bin_hist(Xv, N, S)
    q = eng.normalise(cv)
    qh = q.cpu().numpy()
    assert np.array_equal(qh, onp.normalise(onp.expected_s1(xv, S)))
    o32, o64 = eng.score_s1_from_binhist(Hv, N, S, q, want32=True, want64=True)
    ref = onp.score_s1(xv, qh, S)
    np.testing.assert_allclose(o64.cpu().numpy(), ref, rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(o32.cpu().numpy(), ref.astype(np.float32), rtol=2e-7, atol=0)
    d32, _ = eng.score_s1(Xv, N, S, q)                           # the matrix-scanning route
    assert torch.equal(d32, o32)
    if N >= 2:
        c2 = eng.hist_s2_from_binhist(Hv, S)
        w2 = onp.expected_s2(xv, S)
        assert np.array_equal(c2.cpu().numpy().reshape(S, S), w2)
        Hf, c2f = eng.bin_hist_s2(Xv, N, S)                      # the count pass with the pair counts folded in (or its fallback)
        assert torch.equal(Hf, Hv) and torch.equal(c2f, c2)
        q2 = eng.normalise(c2)
        s32, s64 = eng.score_s2_from_binhist(Hv, N, S, q2, want32=True, want64=True)
        r2 = onp.score_s2(xv, q2.cpu().numpy().reshape(S, S), S)
        np.testing.assert_allclose(s64.cpu().numpy(), r2, rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(s32.cpu().numpy(), r2.astype(np.float32), rtol=3e-7, atol=1e-12)


@pytest.mark.parametrize("S,N,R,seed", _cases(20, 9))
def test_paired_s1_random_shapes(eng, S, N, R, seed):
    """Both groups' histograms -> null groups (a permutation's bookkeeping) -> the one-pass kernel, single and multi-part, against
    the oracle's deltas, STEP 4 reduction and quiescence mask."""
    from epilogos_amd.scores import s1ScoreTable
    NA, NB = N, max(2, (N * 7) // 9)
    xa, xb = synth_states(R, NA, S=S, seed=seed), synth_states(R, NB, S=S, seed=seed + 1)
    if R > 4:
        xa[1:3, :] = S - 1
        xb[1:3, :] = S - 1
    HA, cA = eng.bin_hist(eng.states_to_device(xa), NA, S)
    HB, _ = eng.bin_hist(eng.states_to_device(xb), NB, S, counts=cA)
    q = eng.normalise(cA).cpu().numpy()
    OA, OB = eng.null_hist_from_binhist(HA, HB, NA + NB, S, NA, NB, seed=seed, row0=(3 << 40) + 17)
    try:                                                         # count pass + null draw in one kernel: the same bits where it applies
        cF = torch.zeros(S, dtype=torch.int64, device="cuda")
        fa, fb, foa, fob = eng.pair_count_null_parts([eng.states_to_device(xa)], [eng.states_to_device(xb)], NA, NB, S, seed, [(3 << 40) + 17], counts=cF)
        assert torch.equal(fa[0], HA) and torch.equal(fb[0], HB) and torch.equal(foa[0], OA) and torch.equal(fob[0], OB) and torch.equal(cF, cA)
    except eng.EpilogosHipError as e:
        assert e.code == -2 and (S not in (15, 18, 25) or (NA + 127) // 128 != (NB + 127) // 128)
    tot = eng.hist_to_numpy(HA).astype(np.int64) + eng.hist_to_numpy(HB).astype(np.int64)
    oa, ob = eng.hist_to_numpy(OA).astype(np.int64), eng.hist_to_numpy(OB).astype(np.int64)
    assert np.array_equal(oa + ob, tot) and (oa.sum(axis=1) == NA).all() and (ob.sum(axis=1) == NB).all()
    tA, tB = (torch.from_numpy(s1ScoreTable(q, n)[1]).cuda() for n in (NA, NB))
    try:
        d, n, rd, md = eng.pair_scores_s1_from_binhist(HA, HB, OA, OB, S, NA, NB, NA, NB, tA, tB, tA, tB)
    except eng.EpilogosHipError as e:
        assert e.code == -2                                      # tables too wide for the LDS: the separate passes are the route
        return
    want, _ = onp.pair_finish(onp.score_s1(xa, q, S).astype(np.float32), onp.score_s1(xb, q, S).astype(np.float32))
    assert np.array_equal(d.cpu().numpy(), want)
    wd, wm = onp.pair_metrics(want, True)
    assert np.array_equal(rd.cpu().numpy(), wd) and np.array_equal(md.cpu().numpy(), wm)
    cut = R // 3
    quads = [tuple(t[a:b].clone() for t in (HA, HB, OA, OB)) for a, b in ((0, cut), (cut, cut), (cut, R))]
    res = eng.pair_scores_s1_parts(quads, S, NA, NB, NA, NB, tA, tB, tA, tB, qstate=S - 1)
    assert torch.equal(torch.cat([r["delta"] for r in res]), d) and torch.equal(torch.cat([r["null"] for r in res]), n)
    assert np.array_equal(torch.cat([r["quies"] for r in res]).cpu().numpy().astype(bool), onp.quiescent_mask(xa, xb, S - 1))


def _driver_seeds():
    import os
    n = int(os.environ.get("EPG_RANDOM_DRIVER", 6))
    return list(range(1, n + 1))


@pytest.mark.parametrize("seed", _driver_seeds())
def test_driver_random_inputs_against_oracle(tmp_path, seed):
    """The genome driver on random inputs -- 1 to 4 files of 0 .. 700 rows (empty files and one-row files included), 2 .. 70
    biosamples, 15 / 18 / 25 states, saliency 1 or 2, single or paired -- against the oracle on the concatenated matrix: exp_freq
    exactly, every score of every file to the text's five decimals."""
    import gzip
    from epilogos_amd import driver
    from tests.test_host_logic import write_tsv
    rng = np.random.default_rng(seed)
    S = int(rng.choice([15, 18, 25]))
    N = int(rng.integers(2, 71))
    sal = int(rng.choice([1, 2]))
    paired = bool(rng.integers(0, 2))
    nfiles = int(rng.integers(1, 5))
    rows = [int(rng.choice([0, 1, 63, 64, 65, 200, 511, 700])) for _ in range(nfiles)]
    if sum(rows) == 0:
        rows[0] = 77
    out = tmp_path / "out"
    out.mkdir()
    dirs = [tmp_path / "A", tmp_path / "B"] if paired else [tmp_path / "A"]
    mats = []
    for gi, dpath in enumerate(dirs):
        dpath.mkdir()
        n_g = N if gi == 0 else max(2, N - 3)
        xs = []
        for k, R in enumerate(rows):
            x = synth_states(R, n_g, S=S, seed=seed * 100 + gi * 10 + k, uniform=bool(k % 2)) if R else np.zeros((0, n_g), dtype=np.int8)
            if R == 0:
                (dpath / ("m_chr%d.txt" % (k + 1))).write_text("")
            else:
                write_tsv(dpath / ("m_chr%d.txt" % (k + 1)), x, chrom="chr%d" % (k + 1))
            xs.append(x)
        mats.append(xs)
    files = [sorted(d.glob("*")) for d in dirs]

    def text(name):
        with gzip.open(out / name, "rb") as fh:
            t = fh.read().decode()
        return np.array([[float(v) for v in l.split("\t")[3:]] for l in t.splitlines()], dtype=np.float64).reshape(-1, S)

    if not paired:
        q, _res = driver.run_single_group(files[0], S, sal, out, "t")
        cat = np.concatenate(mats[0])
        want_q = onp.normalise(onp.expected_s1(cat, S) if sal == 1 else onp.expected_s2(cat, S))
        assert np.array_equal(q, want_q)
        for k, x in enumerate(mats[0]):
            got = text("scores_t_m_chr%d.txt.gz" % (k + 1))
            ref = onp.score_s1(x, want_q, S) if sal == 1 else onp.score_s2(x, want_q, S)
            assert got.shape == ref.shape
            np.testing.assert_allclose(got, ref, atol=1.01e-5)
    else:
        q, _res = driver.run_paired_groups(files[0], files[1], S, sal, out, "t", S - 1, -1, 99)
        cat = np.concatenate([np.concatenate([a, b], axis=1) for a, b in zip(mats[0], mats[1])])
        want_q = onp.normalise(onp.expected_s1(cat, S) if sal == 1 else onp.expected_s2(cat, S))
        assert np.array_equal(q, want_q)
        for k, (xa, xb) in enumerate(zip(mats[0], mats[1])):
            got = text("pairwiseDelta_t_m_chr%d.txt.gz" % (k + 1))
            if sal == 1:
                sa, sb = onp.score_s1(xa, want_q, S), onp.score_s1(xb, want_q, S)
            else:
                sa = onp.score_s2(xa, want_q, S, perms=xa.shape[1] * (xa.shape[1] - 1))
                sb = onp.score_s2(xb, want_q, S, perms=xb.shape[1] * (xb.shape[1] - 1))
            ref, _ = onp.pair_finish(sa.astype(np.float32), sb.astype(np.float32))
            assert got.shape == ref.shape
            np.testing.assert_allclose(got, ref, atol=2.01e-5)
            qm = np.load(out / ("temp_quiescence_t_m_chr%d.npz" % (k + 1)))["quiescenceArr"]
            assert np.array_equal(qm, onp.quiescent_mask(xa, xb, S - 1))

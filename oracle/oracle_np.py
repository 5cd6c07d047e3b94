"""
ORACLE (test infrastructure, NOT product code) -- numpy restatement of the epilogos scoring hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
path (epilogos_amd/) never imports it and has no CPU fallback.

Every function cites the reference lines it restates (paths relative to /root/reference/epilogos/).  The
arithmetic specification followed is SURVEY.md Appendix A.

Parity status: PINNED.  tests/golden/*.npz were produced by tests/golden/make_golden.py, which imports the
real reference (numpy 2.2.6 in the build container) and runs its expected.*Calc / scores.*Score / writeScores on
small inputs; tests/test_oracle_golden.py checks this restatement against every one of those vectors
(integers bit-exact, S1/S2 float64 bit-exact on the generating machine and <=1e-12 rel elsewhere).

Conventions: x is int array [R, N] of 0-based states (file value - 1, helpers.py:154-155); S = numStates.
"""
import numpy as np

__all__ = [
    "split_rows", "bin_hist", "expected_s1", "expected_s2", "expected_s3", "normalise", "kl",
    "score_s1", "score_s2", "score_s3_f32_sequential", "score_s3_f64", "s3_table", "pair_finish", "pair_metrics", "text_roundtrip_f5",
    "quiescent_mask", "format_scores", "shuffle_rows",
]


def split_rows(total_rows, num_parts):
    """helpers.py:102-120 -- contiguous ranges (i*R//P, (i+1)*R//P); also the GPU bin-range partition rule."""
    return [(i * total_rows // num_parts, (i + 1) * total_rows // num_parts) for i in range(num_parts)]


def bin_hist(x, S):
    """h[b, s] = #{n : x[b, n] == s}; what np.unique(dataArr[row], return_counts=True) yields per row
    (scores.py:341, scores.py:444, expected.py:152).  States outside [0, S) are not counted."""
    x = np.asarray(x)
    R, N = x.shape
    h = np.zeros((R, S), dtype=np.int64)
    for s in range(S):
        h[:, s] = (x == s).sum(axis=1)
    return h


def expected_s1(x, S):
    """expected.py:106-113 -- int64[S] state counts over the whole chunk."""
    return bin_hist(x, S).sum(axis=0).astype(np.int64)


def expected_s2(x, S):
    """expected.py:137,146-158 -- C[i,j] = sum_b h_i*h_j (i != j), h_i*(h_i-1) (i == j); int64[S,S]."""
    h = bin_hist(x, S)
    C = h.T @ h
    C[np.arange(S), np.arange(S)] -= h.sum(axis=0)
    return C.astype(np.int64)


def expected_s3(x, S, chunk=16384):
    """expected.py:183-200 -- C[a,b,i,j] = #{bins: x[.,a]==i and x[.,b]==j}, a != b; diagonal a == b stays 0;
    int32[N,N,S,S].  Restated as the Gram matrix of the one-hot expansion E[bin, (sample, state)]: 0/1 products summed
    over at most `chunk` < 2^24 bins are exact in float32, chunks are added up in int64.  Only the upper triangle is
    computed (BLAS syrk) and mirrored -- G is symmetric by construction; at N = 833 (14 994 rows) this is what makes the
    full-size check affordable on host cores."""
    from scipy.linalg import blas
    x = np.asarray(x)
    R, N = x.shape
    NS = N * S
    assert chunk < (1 << 24)
    G = np.zeros((NS, NS), dtype=np.int64)
    for lo in range(0, R, chunk):
        xc = x[lo:lo + chunk]
        n = xc.shape[0]
        ET = np.zeros((NS, n), dtype=np.float32)            # E transposed, C order == E in Fortran order (what syrk wants)
        rows = np.repeat(np.arange(n), N)
        cols = (np.arange(N)[None, :] * S + xc).reshape(-1)
        valid = ((xc >= 0) & (xc < S)).reshape(-1)
        ET[cols[valid], rows[valid]] = 1.0
        G += blas.ssyrk(1.0, ET.T, trans=1).astype(np.int64)  # upper triangle of E^T E, exact integers
    G = np.triu(G) + np.triu(G, 1).T
    C = G.reshape(N, S, N, S).transpose(0, 2, 1, 3).astype(np.int32)
    C[np.arange(N), np.arange(N)] = 0
    return C


def normalise(C):
    """expectedCombination.py:42 -- (C / np.sum(C)).astype(float32): float64 true division, then float32 round."""
    C = np.asarray(C)
    return (C / np.sum(C)).astype(np.float32)


def kl(obs, exp):
    """scores.py:550 klScoreND -- obs * log2(obs/exp) with numpy.ma domain semantics:
    obs/exp masked where exp == 0 (-> 0), log2 masked where ratio <= 0 (-> 0).  Result dtype follows numpy
    promotion of obs with exp (float64 if obs is float64; float32 if both are float32)."""
    obs = np.asarray(obs)
    exp = np.asarray(exp)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(exp != 0, np.divide(obs, exp), 0).astype(np.result_type(obs, exp))
        lg = np.where(ratio > 0, np.log2(np.where(ratio > 0, ratio, 1)), 0).astype(ratio.dtype)
        return obs * lg


def score_s1(x, q, S, n_cols=None):
    """scores.py:317,327-344 -- p[s] = h[s]/N (N = width of that array), score = kl(p, q); returns the float64
    value BEFORE the reference's float32 store.  `.astype(float32)` gives the as-stored value."""
    x = np.asarray(x)
    N = x.shape[1] if n_cols is None else n_cols
    p = bin_hist(x, S) / N
    return kl(p, np.asarray(q)[None, :])


def _obs_s2(h, perms):
    """scores.py:443-451 rowObsS2 for all rows: integer numerator, then true division by `permutations`."""
    num = h[:, :, None] * h[:, None, :]
    S = h.shape[1]
    idx = np.arange(S)
    num[:, idx, idx] = h * (h - 1)
    return num / perms


def score_s2(x, q, S, perms=None):
    """scores.py:371,412,426-452 -- p[i,j] = (h_i*h_j - [i==j]*h_i)/P, score[j] = sum_i kl(p, q)[i, j] summed in
    ascending i (numpy .sum(axis=0) over a small non-contiguous axis is a sequential add); float64 pre-store."""
    x = np.asarray(x)
    N = x.shape[1]
    if perms is None:
        perms = N * (N - 1)
    h = bin_hist(x, S)
    out = np.zeros((x.shape[0], S), dtype=np.float64)
    step = 4096
    for r0 in range(0, x.shape[0], step):
        t = kl(_obs_s2(h[r0:r0 + step], perms), np.asarray(q)[None, :, :])
        acc = np.zeros((t.shape[0], S), dtype=np.float64)
        for i in range(S):
            acc = acc + t[:, i, :]
        out[r0:r0 + step] = acc
    return out


def s3_table(q, N, correctly_rounded=False):
    """scores.py:479-480 -- T = klScoreND(ones(float32)/(N*(N-1)), q) in float32, by numpy as the reference runs it.
    correctly_rounded=True: the same float32 expression with every operation correctly rounded (quotient, log2 evaluated in
    float64 and rounded once, product) -- what the device builds (csrc/epg_common.h s3_table_entry).  numpy's own float32 log2 is
    a SIMD routine that is 1-2 ulp off that in a share of its arguments which depends on the host's CPU (tests/test_oracle_golden.py
    puts a number on it); the two tables give scores that agree to < 1e-7."""
    q = np.asarray(q, dtype=np.float32)
    obs = (np.ones(q.shape, dtype=np.float32) / (N * (N - 1))).astype(np.float32)
    if not correctly_rounded:
        return kl(obs, q).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(q != 0, np.divide(obs, q), 0).astype(np.float32)
        lg = np.where(ratio > 0, np.log2(np.where(ratio > 0, ratio, 1).astype(np.float64)), 0).astype(np.float32)
    return (obs * lg).astype(np.float32)


def score_s3_f32_sequential(x, q, S):
    """scores.py:474,487,496-504 -- float32 accumulator, np.add.at order = itertools.permutations(range(N), 2)
    (lexicographic (a, b), a != b): score[x[b]] += T[a, b, x[a], x[b]].  Pure-Python-speed; small cases only."""
    x = np.asarray(x)
    R, N = x.shape
    T = s3_table(q, N)
    a_idx, b_idx = np.nonzero(~np.eye(N, dtype=bool))  # lexicographic ordered pairs a != b
    out = np.zeros((R, S), dtype=np.float32)
    row = np.zeros(S, dtype=np.float32)
    for r in range(R):
        row.fill(0)
        np.add.at(row, x[r, b_idx], T[a_idx, b_idx, x[r, a_idx], x[r, b_idx]])
        out[r] = row
    return out


def score_s3_f64(x, q, S, correctly_rounded=False):
    """Closed form of scores.py:496-498 with a float64 accumulator: score[b, s] = sum_{beta: x_beta == s}
    sum_{alpha != beta} T[alpha, beta, x_alpha, s], T the float32 table promoted to float64."""
    x = np.asarray(x)
    R, N = x.shape
    T = s3_table(q, N, correctly_rounded).astype(np.float64)
    a_idx, b_idx = np.nonzero(~np.eye(N, dtype=bool))
    out = np.zeros((R, S), dtype=np.float64)
    for r in range(R):
        np.add.at(out[r], x[r, b_idx], T[a_idx, b_idx, x[r, a_idx], x[r, b_idx]])
    return out


def pair_finish(score_a, score_b):
    """scores.py:223-232 -- delta = A - B (float32); signed squared distance = sum_s d^2 * sign(sum_s d)."""
    a = np.asarray(score_a, dtype=np.float32)
    b = np.asarray(score_b, dtype=np.float32)
    d = a - b
    dist = np.sum(np.square(d), axis=1) * np.sign(np.sum(d, axis=1))
    return d, dist


def text_roundtrip_f5(v):
    """float32 -> '{:.5f}' (scores.py:530-531) -> pandas read_table -> float32 (roiAndVisualPairwise.py:340), restated
    arithmetically: v * 1e5 is exact in float64, rint is half-even like the decimal formatter, k / 1e5 is the
    correctly rounded double of the decimal string."""
    v = np.asarray(v, dtype=np.float32)
    return (np.rint(v.astype(np.float64) * 1e5) / 1e5).astype(np.float32)


def pair_metrics(delta, roundtrip=True):
    """roiAndVisualPairwise.py:347-354 -- signed squared euclidean distance and the 1-based state of the largest
    |difference| (ties to the higher state) of every bin.  The reference sums a column-major float32 frame over
    axis 1, which numpy does column after column: plain ascending-state float32 adds, no pairwise blocking."""
    d = np.asarray(delta, dtype=np.float32)
    if roundtrip:
        d = text_roundtrip_f5(d)
    sq = np.zeros(d.shape[0], dtype=np.float32)
    sd = np.zeros(d.shape[0], dtype=np.float32)
    for s in range(d.shape[1]):
        sq = sq + d[:, s] * d[:, s]
        sd = sd + d[:, s]
    dist = sq * np.sign(sd)
    maxdiff = np.abs(np.argmax(np.abs(np.flip(d, axis=1)), axis=1) - d.shape[1]).astype(np.int32)
    return dist, maxdiff


def quiescent_mask(xa, xb, qstate):
    """scores.py:294-303 -- bin is quiescent iff every state in A and in B equals qstate (qstate == -1: off)."""
    xa = np.asarray(xa)
    xb = np.asarray(xb)
    if qstate == -1:
        return np.zeros(xa.shape[0], dtype=bool)
    return np.all(xa == qstate, axis=1) & np.all(xb == qstate, axis=1)


def shuffle_rows(combined, rand):
    """helpers.py:183-184 -- per-row permutation by argsort of a uniform matrix `rand` of the same shape."""
    idx = np.argsort(rand, axis=1)
    return np.take_along_axis(combined, idx, axis=1)


def format_scores(loc, scores):
    """scores.py:530-532 -- 'chr\\tstart\\tend\\t' + tab-joined '{:.5f}' of the float32 row + '\\n'."""
    scores = np.asarray(scores, dtype=np.float32)
    lines = []
    for i in range(scores.shape[0]):
        lines.append("{}\t{}\t{}\t".format(loc[i][0], loc[i][1], loc[i][2])
                     + "\t".join("{:.5f}".format(v) for v in scores[i]) + "\n")
    return "".join(lines)

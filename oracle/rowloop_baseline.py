"""
ORACLE / BASELINE (test infrastructure, NOT product code) -- per-bin CPU baseline with the algorithmic shape of the
reference's `-l` S1 score loop: for every bin a sort-based unique-count histogram, p = count / N, and a masked
p * log2(p / q) (numpy.ma), exactly one Python iteration per bin, row ranges fanned out over a process pool.
It stands in for scores.py:309-317 (loop), 327-344 (rowObsS1) and 539-550 (klScoreND) on machines where the
reference itself is not present (the GPU box); SURVEY.md 8d calibrates it against the real reference
(10.3 k bins/s/core at N = 833 on a 2.1 GHz Xeon).

Used only by bench.py's cpu_baseline leg and tests/.
"""
import os
import time
from multiprocessing import get_context

import numpy as np
import numpy.ma as ma

_G = {}


def _kl_masked(obs, exp):
    quotient = ma.divide(obs, exp).filled(0)
    return obs * ma.log2(quotient).filled(0)


def score_rows_s1(x, q, S):
    """One Python iteration per bin, float32 result array like the reference's shared score array."""
    R, N = x.shape
    out = np.zeros((R, S), dtype=np.float32)
    for r in range(R):
        states, cnts = np.unique(x[r], return_counts=True)
        obs = np.zeros(S)
        for k, st in enumerate(states):
            obs[st] = cnts[k] / N
        out[r] = _kl_masked(obs, q)
    return out


def _worker(args):
    lo, hi, deadline = args
    xs = _G["x"][lo:hi].astype(np.int64)          # the reference holds states as int64 (helpers.py:154-155)
    q, S = _G["q"], _G["S"]
    done, acc, block = 0, 0.0, 500
    t0 = time.perf_counter()
    while True:                                    # sweep the slice (again and again) until the deadline
        for r0 in range(0, xs.shape[0], block):
            acc += float(score_rows_s1(xs[r0:r0 + block], q, S).sum())
            done += min(block, xs.shape[0] - r0)
            if time.time() >= deadline:
                return done, time.perf_counter() - t0, acc
        if xs.shape[0] == 0:
            return done, time.perf_counter() - t0, acc


def _noop(_):
    return 0


def timed_pool_run(x, q, S, cores=None, seconds=10.0):
    """Score rows of x with `cores` forked workers over contiguous row ranges (the reference's splitRows rule,
    helpers.py:116-118) for about `seconds` of wall time: every worker keeps sweeping its range until the common
    deadline and reports how many bins it finished.  The pool is started (and warmed) before the clock.
    Returns (bins_per_second, seconds, cores, bins_scored)."""
    cores = cores or len(os.sched_getaffinity(0))
    R = x.shape[0]
    _G.update(x=x, q=q, S=S)
    ctx = get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_noop, range(cores))
        t0 = time.perf_counter()
        deadline = time.time() + seconds
        res = pool.map(_worker, [(i * R // cores, (i + 1) * R // cores, deadline) for i in range(cores)], chunksize=1)
        dt = time.perf_counter() - t0
    bins = sum(r[0] for r in res)
    return bins / dt, dt, cores, bins

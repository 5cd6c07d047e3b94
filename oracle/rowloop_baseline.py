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


def _worker(rng):
    lo, hi = rng
    xs = _G["x"][lo:hi].astype(np.int64)          # the reference holds states as int64 (helpers.py:154-155)
    t0 = time.perf_counter()
    acc = 0.0
    for _ in range(_G["reps"]):
        acc += float(score_rows_s1(xs, _G["q"], _G["S"]).sum())
    return acc, time.perf_counter() - t0


def _noop(_):
    return 0


def timed_pool_run(x, q, S, cores=None, reps=1):
    """Score all rows of x `reps` times with `cores` forked workers over contiguous row ranges (the reference's
    splitRows rule, helpers.py:116-118).  The pool is started (and warmed) before the clock.  Returns
    (bins_per_second, seconds, cores)."""
    cores = cores or len(os.sched_getaffinity(0))
    R = x.shape[0]
    _G.update(x=x, q=q, S=S, reps=reps)
    ranges = [(i * R // cores, (i + 1) * R // cores) for i in range(cores)]
    ctx = get_context("fork")
    with ctx.Pool(cores) as pool:
        pool.map(_noop, range(cores))
        t0 = time.perf_counter()
        pool.map(_worker, ranges, chunksize=1)
        dt = time.perf_counter() - t0
    return R * reps / dt, dt, cores

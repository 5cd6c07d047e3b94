#!/usr/bin/env python3
"""GPU box: largest deviation of the S3 scores (epg_score_s3, float64 output) from the float64 restatement of the reference
(oracle.score_s3_f64: numpy's float32 table, float64 sum) over the shapes of the GPU tests -- relative (cells above 1e-9) and as the
rtol an assert_allclose with atol 1e-9 needs.  Run once per library: EPILOGOS_HIP_LIB=tools/_ab_libs/log2f.so for rounds 1-5's table."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import oracle_np as onp
from tests.conftest import synth_states, load_golden
from epilogos_amd import engine
engine.require_gpu()


def probe(name, x, q, S, rows=None):
    N = x.shape[1]
    X = engine.states_to_device(x)
    qd = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32).reshape(-1)).cuda()
    _, o64 = engine.score_s3(X, N, S, qd, want32=False, want64=True)
    got = o64.cpu().numpy()
    rows = np.arange(min(x.shape[0], 300)) if rows is None else rows
    ref = onp.score_s3_f64(x[rows], q, S)
    err = np.abs(got[rows] - ref)
    big = np.abs(ref) > 1e-9
    rel = float((err[big] / np.abs(ref[big])).max()) if big.any() else 0.0
    need = float((np.maximum(err - 1e-9, 0) / (np.abs(ref) + 1e-300)).max())
    print("%-26s max rel %.2e   rtol needed at atol 1e-9: %.2e" % (name, rel, need), flush=True)
    return need


worst = 0.0
gs, gr, ge = load_golden("s3_small.npz"), load_golden("real_slice.npz"), load_golden("edge.npz")
worst = max(worst, probe("golden s3_small", gs["x"], gs["s3_exp"], 18))
worst = max(worst, probe("golden real_slice[:600]", gr["x"][:600], gr["s3_exp"], 18, np.arange(600)))
worst = max(worst, probe("golden edge N=2", ge["n2_x"], ge["n2_s3_exp"], 18))
for N in (33, 65):
    x = synth_states(211, N, seed=N)
    worst = max(worst, probe("N=%d R=211" % N, x, onp.normalise(onp.expected_s3(x, 18)), 18))
for S_, N, R in [(5, 40, 300), (13, 23, 1000), (15, 64, 700), (25, 30, 257), (30, 21, 420), (31, 9, 300), (15, 200, 1500), (20, 70, 1500), (19, 33, 2900), (21, 40, 300)]:
    x = synth_states(R, N, S=S_, seed=S_, uniform=True)
    worst = max(worst, probe("S=%d N=%d R=%d" % (S_, N, R), x, onp.normalise(onp.expected_s3(x, S_)), S_))
x = synth_states(9000, 70, seed=3)
worst = max(worst, probe("N=70 R=9000", x, onp.normalise(onp.expected_s3(x, 18)), 18))
if "--n833" in sys.argv:
    N, R = 833, 4400
    x = synth_states(R, N, seed=833)
    c = engine.hist_s3(engine.states_to_device(x), N, 18)
    q = engine.normalise(c).cpu().numpy().reshape(N, N, 18, 18)
    worst = max(worst, probe("N=833 R=4400 (24 rows)", x, q, 18, np.arange(0, R, 190)))
print("library %s: worst rtol needed %.2e" % (os.environ.get("EPILOGOS_HIP_LIB", "in-tree"), worst))

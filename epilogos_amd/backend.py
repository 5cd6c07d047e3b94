"""Compute backend used by the stage drivers (expected.py / scores.py / driver.py).

There is exactly one product backend: HipBackend (the gfx950 kernels through the C ABI).  It raises when the HIP
library or a GPU is missing -- there is no CPU fallback.  `set_for_testing` exists so that the CPU-only test-suite
can exercise the host logic (file naming, partitioning, the gloo collective path) with an oracle-backed stand-in;
nothing in the package ever installs one.
"""
import numpy as np

_override = None


class HipBackend:
    """Host-array facade over epilogos_amd.engine (device tensors inside)."""
    name = "hip"

    def __init__(self, device=None):
        import torch
        from . import engine
        engine.require_gpu()
        self.torch = torch
        self.engine = engine
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device

    # ---- transfers
    def to_device(self, x):
        return self.engine.states_to_device(x, device=self.device)

    # ---- expected pass (STEP 1): integer counts as host arrays with the reference's dtypes/shapes
    def expected_counts(self, x, S, saliency):
        eng = self.engine
        N = x.shape[1]
        X = self.to_device(x)
        if saliency == 1:
            _, c = eng.bin_hist(X, N, S, want_hist=False)
            return c.cpu().numpy()
        if saliency == 2:
            H, _ = eng.bin_hist(X, N, S, want_counts=False)
            return eng.hist_s2_from_binhist(H, S).cpu().numpy().reshape(S, S)
        if saliency == 3:
            return eng.hist_s3(X, N, S).cpu().numpy().reshape(N, N, S, S)
        raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")

    # device-resident accumulation for the genome-wide driver: the S3 count array is 899 MB at N = 833, so it stays in
    # HBM from the first chunk to the all-reduce and the normalisation instead of crossing PCIe once per chunk
    def counts_begin(self, S, saliency, N):
        n = {1: S, 2: S * S, 3: N * N * S * S}[saliency]
        return self.torch.zeros(n, dtype=self.torch.int32 if saliency == 3 else self.torch.int64, device=self.device)

    def counts_add(self, acc, x, S, saliency):
        eng = self.engine
        N = x.shape[1]
        X = self.to_device(x)
        if saliency == 1:
            eng.bin_hist(X, N, S, want_hist=False, counts=acc)
        elif saliency == 2:
            H, _ = eng.bin_hist(X, N, S, want_counts=False)
            eng.hist_s2_from_binhist(H, S, counts=acc)
        elif saliency == 3:
            eng.hist_s3(X, N, S, counts=acc)
        else:
            raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")

    def counts_finish(self, acc, shape):
        """(exp_freq on the device for the score pass, exp_freq as the host array that is saved)."""
        qd = self.engine.normalise(acc)
        return qd, qd.cpu().numpy().reshape(shape)

    def check_counts(self, counts, R, N, saliency):
        """Every state byte must have been counted: a byte outside [0, S) is silently skipped by the kernels."""
        total = int(np.asarray(counts, dtype=np.int64).sum())
        want = R * N if saliency == 1 else R * N * (N - 1)
        if total != want:
            raise ValueError("input contains states outside 1..numStates (counted %d of %d)" % (total, want))

    # ---- combination (STEP 2)
    def normalise(self, counts):
        t = self.torch.from_numpy(np.ascontiguousarray(counts).reshape(-1)).to(self.device)
        return self.engine.normalise(t).cpu().numpy().reshape(np.shape(counts))

    # ---- score pass (STEP 3): float32 [R, S] exactly as the reference stores it
    def scores(self, x, S, saliency, q, perms=None):
        eng = self.engine
        N = x.shape[1]
        X = self.to_device(x)
        qd = q.reshape(-1) if self.torch.is_tensor(q) else \
            self.torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32).reshape(-1)).to(self.device)
        if saliency == 1:
            o32, _ = eng.score_s1(X, N, S, qd)
        elif saliency == 2:
            o32, _ = eng.score_s2(X, N, S, qd, perms=perms)
        elif saliency == 3:
            o32, _ = eng.score_s3(X, N, S, qd)
        else:
            raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")
        return o32.cpu().numpy()

    # ---- paired extras
    def pair_finish(self, a, b):
        ta = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        tb = self.torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(self.device)
        d, dist = self.engine.pair_finish(ta, tb)
        return d.cpu().numpy(), dist.cpu().numpy()

    def pair_metrics(self, delta, roundtrip=True):
        td = self.torch.from_numpy(np.ascontiguousarray(delta, dtype=np.float32)).to(self.device)
        dist, maxdiff = self.engine.pair_metrics(td, roundtrip)
        return dist.cpu().numpy(), maxdiff.cpu().numpy()

    def quiescent(self, xa, xb, qstate):
        m = self.engine.quiescent(self.to_device(xa), xa.shape[1], self.to_device(xb), xb.shape[1], qstate)
        return m.cpu().numpy().astype(bool)

    def null_scores(self, xa, xb, S, saliency, q, groupSize, seed, row0=0):
        """Scores of the two shuffled null halves (reference helpers.py:183-194 + scores.py:321-322,418-421)."""
        eng = self.engine
        NA, NB = xa.shape[1], xb.shape[1]
        ga, gb = (NA, NB) if groupSize == -1 else (groupSize, groupSize)
        HA, HB = eng.null_hist(self.to_device(xa), NA, self.to_device(xb), NB, S, ga, gb, seed, row0)
        qd = self.torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32).reshape(-1)).to(self.device)
        if saliency == 1:
            na, _ = eng.score_s1_from_binhist(HA, ga, S, qd)
            nb, _ = eng.score_s1_from_binhist(HB, gb, S, qd)
        elif saliency == 2:
            # quirk Q9: the null halves keep the ORIGINAL groups' permutation counts (scores.py:397-398,418-421)
            na, _ = eng.score_s2_from_binhist(HA, max(ga, NA), S, qd, perms=NA * (NA - 1))
            nb, _ = eng.score_s2_from_binhist(HB, max(gb, NB), S, qd, perms=NB * (NB - 1))
        else:
            raise ValueError("Please ensure that saliency metric is either 1 or 2 for Pairwise Epilogos")
        return na.cpu().numpy(), nb.cpu().numpy()


def get():
    if _override is not None:
        return _override
    return HipBackend()


def set_for_testing(obj):
    """tests/ only."""
    global _override
    _override = obj

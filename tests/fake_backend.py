"""Oracle-backed stand-in for epilogos_amd.backend.HipBackend -- tests only (CPU host-logic and gloo tests)."""
import numpy as np

from oracle import oracle_np as onp


class OracleBackend:
    name = "oracle-for-tests"

    def expected_counts(self, x, S, saliency):
        return {1: onp.expected_s1, 2: onp.expected_s2, 3: onp.expected_s3}[saliency](x, S)

    def check_counts(self, counts, R, N, saliency):
        total = int(np.asarray(counts, dtype=np.int64).sum())
        want = R * N if saliency == 1 else R * N * (N - 1)
        if total != want:
            raise ValueError("input contains states outside 1..numStates (counted %d of %d)" % (total, want))

    def normalise(self, counts):
        return onp.normalise(counts)

    def scores(self, x, S, saliency, q, perms=None):
        if saliency == 1:
            return onp.score_s1(x, q, S).astype(np.float32)
        if saliency == 2:
            return onp.score_s2(x, q, S, perms=perms).astype(np.float32)
        return onp.score_s3_f64(x, q, S).astype(np.float32)

    def pair_finish(self, a, b):
        return onp.pair_finish(a, b)

    def pair_metrics(self, delta, roundtrip=True):
        return onp.pair_metrics(delta, roundtrip)

    def quiescent(self, xa, xb, qstate):
        return onp.quiescent_mask(xa, xb, qstate)

    def null_scores(self, xa, xb, S, saliency, q, groupSize, seed, row0=0):
        comb = np.concatenate([xa, xb], axis=1)
        # keyed by (seed, global row) like the engine's Philox shuffle: independent of how bins are split over ranks
        rand = np.stack([np.random.default_rng([seed, row0 + r]).random(comb.shape[1]) for r in range(comb.shape[0])]) \
            if comb.shape[0] else np.zeros(comb.shape)
        sh = onp.shuffle_rows(comb, rand)
        ga, gb = (xa.shape[1], xb.shape[1]) if groupSize == -1 else (groupSize, groupSize)
        na, nb = sh[:, :ga], sh[:, ga:ga + gb]
        if saliency == 1:
            return onp.score_s1(na, q, S).astype(np.float32), onp.score_s1(nb, q, S).astype(np.float32)
        pa, pb = xa.shape[1] * (xa.shape[1] - 1), xb.shape[1] * (xb.shape[1] - 1)
        return (onp.score_s2(na, q, S, perms=pa).astype(np.float32), onp.score_s2(nb, q, S, perms=pb).astype(np.float32))

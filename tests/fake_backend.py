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

    # the driver's session protocol (backend.HipBackend.open_single / open_paired) over host arrays
    def open_single(self, S, saliency):
        return _HostSingleSession(self, S, saliency)

    def open_paired(self, S, saliency, quiescentState, groupSize, seed):
        return _HostPairedSession(self, S, saliency, quiescentState, groupSize, seed)


# ---- sessions over the host-array stand-in: same protocol as the device-resident sessions of backend.HipBackend
# (alloc / skip / add_part / stage / add_staged / set_row0 / ensure_acc / all_reduce / finish / scores / results),
# arithmetic through the stand-in's array methods
class _HostSession:
    def __init__(self, be, S, saliency):
        self.be, self.S, self.sal = be, S, saliency
        self.counts, self.parts, self.q = None, [], None
        self.n_uploads = 0

    def alloc(self, ticket):
        return None

    def skip(self, ticket):
        pass

    def _add_counts(self, c):
        self.counts = c if self.counts is None else self.counts + c

    def all_reduce(self, d):
        self.counts = d.all_reduce_counts(self.counts)

    def launch(self, total_rows, N, pids):
        pass                                           # (the host stand-in has nothing to enqueue ahead of its checks)

    def _finish(self, total_rows, N, shape):
        self.be.check_counts(self.counts, total_rows, N, self.sal)
        self.q = self.be.normalise(self.counts)
        return self.q


class _HostSingleSession(_HostSession):
    def add_part(self, arr, N, ticket):
        x = arr[:, :N]
        self._add_counts(self.be.expected_counts(x, self.S, self.sal))
        self.parts.append(x)
        return len(self.parts) - 1

    n_export = 1

    def slice_part(self, pid, lo, hi, row0=None):
        self.parts.append(self.parts[pid][lo:hi])
        return len(self.parts) - 1

    def export_rows(self, pid, lo, hi):
        return [np.ascontiguousarray(self.parts[pid][lo:hi])]

    def import_rows(self, tensors, N, row0=None):
        self.parts.append(tensors[0].cpu().numpy())
        return len(self.parts) - 1

    def drop_part(self, pid):
        self.parts[pid] = None

    def ensure_acc(self, N):
        if self.counts is None:                        # a rank without bins still takes part in the all-reduce
            S = self.S
            self.counts = np.zeros({1: (S,), 2: (S, S), 3: (N, N, S, S)}[self.sal], dtype=np.int32 if self.sal == 3 else np.int64)

    def finish(self, total_rows, N):
        S = self.S
        return self._finish(total_rows, N, {1: (S,), 2: (S, S), 3: (N, N, S, S)}[self.sal])

    def scores(self, pid):
        x, self.parts[pid] = self.parts[pid], None
        return self.be.scores(x, self.S, self.sal, self.q)


class _HostPairedSession(_HostSession):
    def __init__(self, be, S, saliency, quiescentState, groupSize, seed):
        super().__init__(be, S, saliency)
        self.qstate, self.groupSize, self.seed = quiescentState, groupSize, seed

    def stage(self, arr, N, ticket):
        return arr[:, :N]

    def set_row0(self, pid, row0):
        xa, xb, _ = self.parts[pid]
        self.parts[pid] = (xa, xb, row0)

    def add_part(self, arrA, NA, ticketA, arrB, NB, ticketB, row0):
        return self.add_staged(arrA[:, :NA], NA, arrB[:, :NB], NB, row0)

    def add_staged(self, xa, NA, xb, NB, row0):
        self._add_counts(self.be.expected_counts(np.concatenate((xa, xb), axis=1), self.S, self.sal))
        self.parts.append((xa, xb, row0))
        return len(self.parts) - 1

    n_export = 2

    def slice_part(self, pid, lo, hi, row0=None):
        xa, xb, _ = self.parts[pid]
        self.parts.append((xa[lo:hi], xb[lo:hi], row0))
        return len(self.parts) - 1

    def export_rows(self, pid, lo, hi):
        xa, xb, _ = self.parts[pid]
        return [np.ascontiguousarray(xa[lo:hi]), np.ascontiguousarray(xb[lo:hi])]

    def import_rows(self, tensors, widths, row0=None):
        self.parts.append((tensors[0].cpu().numpy(), tensors[1].cpu().numpy(), row0))
        return len(self.parts) - 1

    def drop_part(self, pid):
        self.parts[pid] = None

    def ensure_acc(self, N):
        if self.counts is None:
            self.counts = np.zeros((self.S,) if self.sal == 1 else (self.S, self.S), dtype=np.int64)

    def finish(self, total_rows, N):
        return self._finish(total_rows, N, (self.S,) if self.sal == 1 else (self.S, self.S))

    def results(self, pid):
        be, S, sal, q = self.be, self.S, self.sal, self.q
        (xa, xb, row0), self.parts[pid] = self.parts[pid], None
        n1, n2 = xa.shape[1], xb.shape[1]
        s1 = be.scores(xa, S, sal, q, perms=n1 * (n1 - 1))
        s2 = be.scores(xb, S, sal, q, perms=n2 * (n2 - 1))
        na, nb = be.null_scores(xa, xb, S, sal, q, self.groupSize, self.seed, row0=row0)
        delta, _ = be.pair_finish(s1, s2)
        _, null = be.pair_finish(na, nb)
        rdist, mdiff = be.pair_metrics(delta, roundtrip=True)      # what STEP 4 would recompute from the text
        return {"delta": delta, "null": null, "quies": be.quiescent(xa, xb, self.qstate), "rdist": rdist, "mdiff": mdiff}

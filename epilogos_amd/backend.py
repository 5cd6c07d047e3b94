"""Compute backend used by the stage drivers (expected.py / scores.py / driver.py).

There is exactly one product backend: HipBackend (the gfx950 kernels through the C ABI).  It raises when the HIP
library or a GPU is missing -- there is no CPU fallback.  `set_for_testing` exists so that the CPU-only test-suite
can exercise the host logic (file naming, partitioning, the gloo collective path) with an oracle-backed stand-in;
nothing in the package ever installs one.
"""
import os

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
            _H, c2 = eng.bin_hist_s2(X, N, S)
            return c2.cpu().numpy().reshape(S, S)
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
            eng.bin_hist_s2(X, N, S, counts2=acc)
        elif saliency == 3:
            eng.hist_s3(X, N, S, counts=acc)
        else:
            raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")

    def counts_finish(self, acc, shape):
        """(exp_freq on the device for the score pass, exp_freq as the host array that is saved)."""
        qd = self.engine.normalise(acc)
        return qd, qd.cpu().numpy().reshape(shape)

    def check_counts(self, counts, R, N, saliency):
        """Every state byte must have been counted: a byte outside [0, S) is silently skipped by the kernels.  `counts` is
        the count array or its already computed total."""
        total = int(counts) if np.ndim(counts) == 0 else int(np.asarray(counts, dtype=np.int64).sum())
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
            o32 = self._score_s1_host_table(eng.bin_hist(X, N, S, want_counts=False)[0], N, S, q)
        elif saliency == 2:
            o32, _ = eng.score_s2(X, N, S, qd, perms=perms)
        elif saliency == 3:
            o32, _ = eng.score_s3(X, N, S, qd)
        else:
            raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")
        return o32.cpu().numpy()

    def _score_s1_host_table(self, H, N, S, q):
        """S1 scores from histograms through the table the host builds with the reference's arithmetic (scores.s1ScoreTable):
        float32 values identical to the reference's, hence identical text."""
        from .scores import s1ScoreTable
        qh = q.cpu().numpy() if self.torch.is_tensor(q) else np.asarray(q, dtype=np.float32)
        _t64, t32 = s1ScoreTable(qh.reshape(-1), N)
        o32, _ = self.engine.score_s1_from_binhist_table(H, N, S, T32=self.torch.from_numpy(t32).to(self.device))
        return o32

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
        if S > 31:                                       # the wide models: the matrix-scanning kernel decodes five bits
            hA, _ = eng.bin_hist(self.to_device(xa), NA, S, want_counts=False)
            hB, _ = eng.bin_hist(self.to_device(xb), NB, S, want_counts=False)
            HA, HB = eng.null_hist_from_binhist(hA, hB, NA + NB, S, ga, gb, seed, row0)
        else:
            HA, HB = eng.null_hist(self.to_device(xa), NA, self.to_device(xb), NB, S, ga, gb, seed, row0)
        qd = self.torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32).reshape(-1)).to(self.device)
        if saliency == 1:
            na, nb = self._score_s1_host_table(HA, ga, S, q), self._score_s1_host_table(HB, gb, S, q)
        elif saliency == 2:
            # quirk Q9: the null halves keep the ORIGINAL groups' permutation counts (scores.py:397-398,418-421)
            na, _ = eng.score_s2_from_binhist(HA, max(ga, NA), S, qd, perms=NA * (NA - 1))
            nb, _ = eng.score_s2_from_binhist(HB, max(gb, NB), S, qd, perms=NB * (NB - 1))
        else:
            raise ValueError("Please ensure that saliency metric is either 1 or 2 for Pairwise Epilogos")
        return na.cpu().numpy(), nb.cpu().numpy()


    # ---- device-resident sessions of the genome-wide driver (driver.py): every part is uploaded ONCE, its per-bin
    # histograms (S1/S2) or its state matrix (S3, paired) stay in HBM between the count pass and the score pass
    def open_single(self, S, saliency):
        return _HipSingleSession(self, S, saliency)

    def open_paired(self, S, saliency, quiescentState, groupSize, seed):
        return _HipPairedSession(self, S, saliency, quiescentState, groupSize, seed)


def _aligned_rows(T, lo, hi):
    """Rows [lo, hi) of a resident part as the kernels want them: the score entry points take 16-byte aligned bases
    (epg_s1.hip check_score_from_hist_args, epg_null.hip, epg_s2.hip), and a row slice of an [R, S] uint16 array starts at
    lo * S * 2 bytes -- a view only when that is a multiple of 16, else a copy of just these rows."""
    t = T[lo:hi]
    if t.data_ptr() % 16 == 0 and t.untyped_storage().nbytes() <= 4 * max(t.numel() * t.element_size(), 1):
        return t
    # a copy of just these rows: a misaligned slice -- or one that would keep alive an allocation several times its size (the
    # histograms of a whole batch of parts come from ONE allocation, engine.hist_rows_alloc: a rank that hands most of a batch
    # over to its neighbours and drops it must not keep all of it for the few rows it scores itself)
    return t.clone()


class _HipSession:
    """Shared plumbing: pinned staging, one asynchronous upload per matrix, the count accumulator."""

    def __init__(self, be, S, saliency):
        self.be, self.S, self.sal = be, S, saliency
        self.torch, self.eng, self.device = be.torch, be.engine, be.device
        self._pool = None                                # page-locked staging (lazily: a session fed device-resident parts has none)
        self._copy_stream = None
        self.held = {}                                   # ticket -> pinned buffer handed to the parser
        self.acc = None
        self.q = None
        self.parts = []
        self.n_uploads = 0
        self.upload_bytes = 0
        self._releaser = None
        self._t1 = {}                                    # S1 score tables (device, float32) by group width
        self._t1_verified = set()
        self.tables_patched = 0
        self._pending_check = None
        self._ws3 = None                                 # S3 expected pass: ONE workspace for all parts of the session
        self._launched = False                           # launch() ran: STEP 2 is on the device, STEP 3 of `_early` parts enqueued
        self._early = {}                                 # part id -> its results, computed before the host-side checks

    @property
    def pool(self):
        """Staging buffers in flight; paired mode holds a part's A and B at once, so never fewer than two (first come, first
        served: the driver takes the parts as their parsers finish).  A reader holds its buffer from the moment it knows its
        file's shape until the upload is over -- the parse runs straight into it -- so fewer buffers than readers serialise the
        parse stage (profiles/r04k: four buffers for sixteen readers, every reader waited 0.3-1.7 s).  Half as many as this rank
        may run parser threads, at most 8 (~1 GB of page-locked memory each for a whole-genome run)."""
        if self._pool is None:
            from . import _io
            n = int(os.environ.get("EPILOGOS_PINNED_BUFFERS", min(8, max(4, _io.host_budget() // 2))))
            self._pool = self.be.engine.PinnedPool(max(2, n), in_order=False)
        return self._pool

    @property
    def copy_stream(self):
        if self._copy_stream is None:
            self._copy_stream = self.torch.cuda.Stream(device=self.device)
        return self._copy_stream

    def alloc(self, ticket):
        """-> alloc(R, N) for _io.read_table / helpers.readTable: a pinned, row-padded destination for part `ticket`."""
        def make(R, N):
            ldx = self.eng.padded_width(N)
            buf = self.held.get(ticket)                  # asked twice for one part (helpers.readTable's pandas re-read of a file
            if buf is None or buf.numel() < R * ldx:     # the native parser refused): the same buffer when it fits
                if buf is not None:
                    self.pool.release(self.held.pop(ticket))
                buf = self.pool.acquire(ticket, R * ldx)
                self.held[ticket] = buf
            return buf[:R * ldx].view(R, ldx).numpy()
        return make

    def skip(self, ticket):
        if ticket not in self.held:
            self.pool.skip(ticket)

    def _upload(self, arr, N, ticket):
        """One H2D copy of the staged matrix (pinned -> HBM on the copy stream).  The device buffer is allocated ON the copy
        stream, so the copy waits for nothing the compute stream is doing (round 2 made it wait for the previous part's
        kernels and then blocked the host until the copy was over); the compute stream waits for the copy's event, and the
        staging buffer goes back to the pool from a helper thread once that event has completed."""
        R = arr.shape[0]
        buf = self.held.pop(ticket, None)
        if buf is None:                                  # not staged through alloc(): pageable fallback of the same layout
            self.pool.skip(ticket)
            X = self.eng.states_to_device(arr[:, :N], device=self.device)
        else:
            X, ev = self.eng.upload_states(buf, R, arr.shape[1], self.copy_stream, device=self.device)
            self.torch.cuda.current_stream().wait_event(ev)
            self._release_later(ev, buf)
        self.n_uploads += 1
        self.upload_bytes += X.numel()
        return X

    def _release_later(self, ev, buf):
        import queue
        import threading
        if self._releaser is None:
            self._release_q = queue.SimpleQueue()

            def work():
                while True:
                    item = self._release_q.get()
                    if item is None:
                        return
                    e, b = item
                    e.synchronize()                      # (releases the GIL) the copy out of this staging buffer is over
                    self.pool.release(b)
            self._releaser = threading.Thread(target=work, name="epilogos-staging-release", daemon=True)
            self._releaser.start()
        self._release_q.put((ev, buf))

    def _acc(self, n, dtype=None):
        if self.acc is None:
            self.acc = self.torch.zeros(n, dtype=dtype or self.torch.int64, device=self.device)
        return self.acc

    def all_reduce(self, d):
        d.all_reduce_tensor(self.acc)                    # RCCL over xGMI: the tensor never leaves HBM

    def _score_s1(self, H, N):
        """S1 score pass: gathers from the [N + 1, S] float32 table of this group width (see _s1_table)."""
        o32, _ = self.eng.score_s1_from_binhist_table(H, N, self.S, T32=self._s1_table(N))
        return o32

    def _s1_widths(self):
        return []

    def _s1_table(self, N):
        """The S1 score table of group width N.  finish_device builds the tables of the session's widths ON THE DEVICE in the
        launch that normalises (k_s1_combine; no host round trip between the all-reduce and the score pass); the host-facing
        finish() then compares them bit for bit with the table numpy builds from the same exp_freq by the reference's own
        expression (scores.s1ScoreTable) and replaces a table that differs -- so scores_*.txt.gz stay the reference's bytes
        whatever the last bit of a device logarithm does.  A width nobody announced is built from the host table."""
        if N not in self._t1:
            from .scores import s1ScoreTable
            _t64, t32 = s1ScoreTable(self.q.cpu().numpy(), N)
            self._t1[N] = self.torch.from_numpy(t32).to(self.device)
        return self._t1[N]

    def verify_tables(self):
        """-> number of device-built S1 tables that had to be replaced by the host's (0 = the device's float32 tables ARE the
        reference's).  Synchronises; called by finish() and, after its timed region, by bench.py."""
        if self.sal != 1 or self.q is None:
            return 0
        from .scores import s1ScoreTable
        qh = self.q.cpu().numpy()
        patched = 0
        for N, Td in list(self._t1.items()):
            if N in self._t1_verified:
                continue
            _t64, t32 = s1ScoreTable(qh, N)
            if not np.array_equal(Td.cpu().numpy().view(np.uint32).reshape(-1), t32.view(np.uint32).reshape(-1)):
                self._t1[N] = self.torch.from_numpy(t32).to(self.device)
                patched += 1
            self._t1_verified.add(N)
        self.tables_patched += patched
        return patched

    def finish_device(self, total_rows, N):
        """STEP 2 on the device, no host synchronisation: exp_freq (flat device tensor) and, for S1, the score tables.  The
        count check (the reference dies on a state outside the model, expected.py:113; here such a byte is counted nowhere, so
        the total comes out short) is DEFERRED to check(), which whoever takes results off the device calls -- finish(),
        scores(), results() do; S3's 899 MB array is summed at once."""
        self._pending_check = (self.acc, total_rows, N)
        if self.sal == 3:
            self.check()
            self.q = self.eng.normalise(self.acc)
        elif self.sal == 1:
            widths = [w for w in self._s1_widths() if w] or [N]
            self.q = None
            for w in dict.fromkeys(widths):
                self.q, _T64, T32 = self.eng.s1_tables(self.acc, w, self.S, q=self.q)
                self._t1[w] = T32
        else:
            self.q = self.eng.normalise(self.acc)
        self.acc = None
        return self.q

    def check(self):
        if self._pending_check is not None:
            acc, total_rows, N = self._pending_check
            self._pending_check = None
            self.be.check_counts(int(acc.sum(dtype=self.torch.int64).item()), total_rows, N, self.sal)

    def launch(self, total_rows, N, pids):
        """Everything of STEP 2 and STEP 3 that needs no host: exp_freq (and the S1 tables) on the device, then the score pass
        of the parts `pids` -- all enqueued, no synchronisation.  finish() follows with the host side (count check, table
        verification, exp_freq as a host array); scores() / results() then hand the parts' results out.  This is the one
        sequence both the command line (driver.run_single / run_paired) and bench.py's step run."""
        self.finish_device(total_rows, N)
        self.launch_scores(pids)

    def launch_scores(self, pids):
        """The second half of launch() (for a caller that wants an event between STEP 2 and STEP 3)."""
        self._launched = True
        self._begin([p for p in pids if p is not None])

    def _finish(self, total_rows, N, shape):
        """The host side of STEP 2: the count check (the reference dies on a state outside the model, expected.py:113), the
        device-built S1 tables against numpy's (a table that differs is replaced and the parts scored before are scored
        again: never seen, tools/s1_table_probe.py), exp_freq as the host array that is saved."""
        if not self._launched:
            self.finish_device(total_rows, N)
        self._launched = False
        self.check()
        if self.verify_tables() and self._early:
            self._begin(list(self._early))               # (the early results came from a table that was replaced)
        self._settle()
        return self.q.cpu().numpy().reshape(shape)


_PENDING = object()                                      # a part whose count pass has not been launched yet (see _HipSingleSession._flush)


class _HipSingleSession(_HipSession):
    def __init__(self, be, S, saliency):
        super().__init__(be, S, saliency)
        self._pending, self._pending_rows = [], 0        # S1: (pid, X, N) of part-sized matrices waiting for their ONE count launch
        self._batches = []                               # S1: {"pids", "flat", "starts", "rows", "N"} of the batches counted so far

    def add_part(self, arr, N, ticket):
        return self.add_device(self._upload(arr, N, ticket), N)

    # S1 parts of less than a GiB -- the chromosome files of a genome -- are COUNTED IN BATCHES: one epg_bin_hist_parts launch per
    # 8 M rows or 32 parts (or when anything needs their histograms), histograms in one flat allocation, and later ONE score
    # launch over that allocation.  Per part this was 24 count launches and 24 score launches of 0.1-0.25 ms with their ramps and
    # tails: 3.42 ms per 15 M-bin genome against 2.6 ms as one matrix (bench.py s1_paths, round 6); in batches the same job
    # is two count launches and two score launches.  Same integers, same float32 scores.
    BATCH_ROWS, BATCH_PARTS = 8_000_000, 32

    def add_device(self, X, N, place=None):
        """Count pass over a RESIDENT part -- the ONE entry of the command line (add_part, after its upload), of bench.py and of
        library callers.  place=None (everybody's default): engine.alloc_hist decides -- the histogram cache of a matrix of a GiB
        or more goes where its search finds another memory class than the matrix's (a process-lifetime home block, one bounded
        probe per device), anything smaller gets a plain allocation, and so does everything when placement is off
        (engine.placement_enabled: EPILOGOS_PLACEMENT=0, ranks sharing a GPU).  The command line's parts are one chromosome file
        each -- under a GiB up to ~880 columns -- so a whole-genome run of the reference's shape counts with plain allocations
        (bench.py reports that figure as placement.unplaced; its count passes hide under the parse anyway) while a caller that holds
        the genome as one matrix gets the placed cache.  place=False forces a plain allocation (bench.py's comparison)."""
        eng, S = self.eng, self.S
        self.N = max(getattr(self, "N", 0) or 0, N or 0)  # (an empty file has no width)
        if X.shape[0] == 0 or not N:                     # an empty part: nothing to count, and the ABI rejects a zero width
            self.parts.append(self.torch.empty((0, S), dtype=self.torch.int16, device=self.device) if self.sal < 3 else X)
            return len(self.parts) - 1
        if self.sal == 1 and place is None and X.numel() < eng.PLACE_MIN_BYTES and os.environ.get("EPILOGOS_SINGLE_BATCH", "1") != "0":
            self.parts.append(_PENDING)
            self._pending.append((len(self.parts) - 1, X, N))
            self._pending_rows += X.shape[0]
            if self._pending_rows >= self.BATCH_ROWS or len(self._pending) >= self.BATCH_PARTS:
                self._flush()
            return len(self.parts) - 1
        if self.sal == 1:
            H, _ = eng.bin_hist(X, N, S, counts=self._acc(S), H=eng.alloc_hist(X, N, S) if place is not False else None)
            self.parts.append(H)
        elif self.sal == 2:                              # the count pass with the pair counts folded in: one launch
            H, _ = eng.bin_hist_s2(X, N, S, counts2=self._acc(S * S), H=eng.alloc_hist(X, N, S) if place is not False else None)
            self.parts.append(H)
        elif self.sal == 3:
            # the ~8 GB workspace of the matrix-core contraction is allocated once per session and grows to the largest
            # part (round 2 allocated and freed one per part: freed device memory is scrubbed at every HBM kernel's expense)
            need = eng.hist_s3_ws_bytes(X.shape[0], N, S)
            if self._ws3 is None or self._ws3.numel() < need:
                self._ws3 = None
                self._ws3 = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
            eng.hist_s3(X, N, S, counts=self._acc(N * N * S * S, self.torch.int32), ws=self._ws3)
            self.parts.append(X)
        else:
            raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")
        return len(self.parts) - 1

    def _flush(self):
        """The count pass of the pending S1 parts: one launch (per schedule class of their widths), histograms from one allocation."""
        if not self._pending:
            return
        eng, S = self.eng, self.S
        batch, self._pending, self._pending_rows = self._pending, [], 0
        rows = [X.shape[0] for _pid, X, _n in batch]
        flat, starts = eng.hist_rows_flat(rows, S, self.device)
        Hs = [flat[a:a + r] for a, r in zip(starts, rows)]
        eng.bin_hist_parts([X for _pid, X, _n in batch], [n for _pid, _X, n in batch], S, counts=self._acc(S), Hs=Hs)
        for (pid, _X, _n), H in zip(batch, Hs):
            self.parts[pid] = H
        if len({n for _pid, _X, n in batch}) == 1:       # (one group width: the batch can be scored in one launch too)
            self._batches.append({"pids": [pid for pid, _X, _n in batch], "flat": flat, "starts": starts, "rows": rows, "N": batch[0][2]})

    # ---- multi-rank hand-over (driver._redistribute): a part is what the score pass reads -- per-bin histograms (S1, S2)
    # or state rows (S3)
    n_export = 1

    def slice_part(self, pid, lo, hi, row0=None):
        self._flush()
        self.parts.append(_aligned_rows(self.parts[pid], lo, hi))
        return len(self.parts) - 1

    def export_rows(self, pid, lo, hi):
        self._flush()
        return [self.parts[pid][lo:hi]]

    def import_rows(self, tensors, N, row0=None):
        self.N = N
        self.parts.append(tensors[0].to(self.device))
        return len(self.parts) - 1

    def drop_part(self, pid):
        self._flush()
        self.parts[pid] = None

    def ensure_acc(self, N):
        self._flush()
        S = self.S                                       # a rank without bins still takes part in the all-reduce
        self.N = N
        self._acc({1: S, 2: S * S, 3: N * N * S * S}[self.sal], self.torch.int32 if self.sal == 3 else None)

    def all_reduce(self, d):
        self._flush()
        super().all_reduce(d)

    def finish_device(self, total_rows, N):
        self._flush()
        return super().finish_device(total_rows, N)

    def finish(self, total_rows, N):
        S = self.S
        self._ws3 = None
        return self._finish(total_rows, N, {1: (S,), 2: (S, S), 3: (N, N, S, S)}[self.sal])

    def scores_device(self, pid, keep=False):
        """float32 [R, S] scores of part `pid` from its resident data, as a device tensor."""
        self._flush()
        eng, S, N = self.eng, self.S, self.N
        D = self.parts[pid]
        if not keep:
            self.parts[pid] = None                       # the part's device data is released with its scores
        if D.shape[0] == 0:
            return self.torch.empty((0, S), dtype=self.torch.float32, device=self.device)
        if self.sal == 1:
            o32 = self._score_s1(D, N)
        elif self.sal == 2:
            o32, _ = eng.score_s2_from_binhist(D, N, S, self.q)
        else:                                            # one score workspace (table, transposed matrix, cells) for all parts
            need = eng.hist_s3_ws_bytes(D.shape[0], N, S)
            if self._ws3 is None or self._ws3.numel() < need:
                self._ws3 = None
                self._ws3 = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
            o32, _ = eng.score_s3(D, N, S, self.q, ws=self._ws3)
        return o32

    def _s1_widths(self):
        return [getattr(self, "N", None)]

    def _begin(self, pids):
        self._flush()
        want = [pid for pid in pids if self.parts[pid] is not None]
        self._early = {}
        asked = set(want)
        for b in self._batches:
            # a batch whose parts are all asked for and still hold the rows the count pass left: ONE score launch over its flat
            # histogram buffer, the parts' scores are views of one flat result
            if self.sal == 1 and b["N"] == self.N and all(pid in asked and self.parts[pid] is not None and self.parts[pid].shape[0] == r
                                                          and self.parts[pid].data_ptr() == b["flat"][a:].data_ptr()
                                                          for pid, a, r in zip(b["pids"], b["starts"], b["rows"])):
                o32 = self._score_s1(b["flat"], self.N)
                for pid, a, r in zip(b["pids"], b["starts"], b["rows"]):
                    self._early[pid] = o32[a:a + r]
        for pid in want:
            if pid not in self._early:
                self._early[pid] = self.scores_device(pid, keep=True)

    def _settle(self):
        for pid in self._early:                          # verified: the resident data of the scored parts can go
            self.parts[pid] = None
        self._batches = [b for b in self._batches if any(self.parts[pid] is not None for pid in b["pids"])]

    def early_scores(self, pid):
        """Device tensor of a part scored by launch() (bench.py: the scores stay in HBM)."""
        return self._early[pid]

    def scores(self, pid):
        self.check()
        if pid in self._early:
            self.parts[pid] = None
            return self._early.pop(pid).cpu().numpy()
        return self.scores_device(pid).cpu().numpy()


class _HipPairedSession(_HipSession):
    def __init__(self, be, S, saliency, quiescentState, groupSize, seed):
        if saliency not in (1, 2):
            raise ValueError("Please ensure that saliency metric is either 1 or 2 for Pairwise Epilogos")
        super().__init__(be, S, saliency)
        self.qstate, self.groupSize, self.seed = quiescentState, groupSize, seed
        self._ready = None                               # results of all parts, computed at the first results() call
        self._pending, self._pending_rows = [], 0        # parts whose count pass has not been launched yet (see add_staged)
        if getattr(be, "_null_stream", None) is None:    # one second stream per backend, not per session (used by the two-kernel path)
            be._null_stream = self.torch.cuda.Stream(device=self.device)
        self.null_stream = be._null_stream

    def stage(self, arr, N, ticket):
        """One group's matrix of a part -> HBM as soon as it is parsed (its staging buffer goes back to the pool at once: a
        part must not sit on two of the few buffers while its other half is still being inflated)."""
        return self._upload(arr, N, ticket)

    def set_row0(self, pid, row0):
        self._flush()
        XA, XB, HA, HB, _, null = self.parts[pid]
        self.parts[pid] = (XA, XB, HA, HB, row0, null)
        if null is None and HA.shape[0]:
            self._start_null([pid])

    def add_part(self, arrA, NA, ticketA, arrB, NB, ticketB, row0):
        return self.add_staged(self._upload(arrA, NA, ticketA), NA, self._upload(arrB, NB, ticketB), NB, row0)

    # a batch of parts is counted in ONE launch (epg_bin_hist_parts over the A and the B matrices of all its parts) and its null
    # groups are drawn in ONE launch on the second stream, under the count pass of the next batch (the sampler is VALU-bound,
    # the count pass HBM-bound: tools/overlap_probe.py).  Round 4 launched per part: 2 x 24 count passes of ~80 us and 24 samplers
    # for a genome, each with its ramp and tail -- the count phase of BASELINE config 5 ran at 0.36 of its bytes.
    # rows per batch: the default group sizes take the fused kernel (nothing to overlap: two launches per genome, 4.62 ms per
    # 15 M bins against 4.71 with seven); the two-kernel path overlaps batch k's sampler with batch k + 1's count pass
    # (1 / 2 / 3 / 4 M rows: 5.04 / 5.05 / 5.12 / 5.10 ms before the score pass prefetched, one batch 6.1)
    BATCH_ROWS = os.environ.get("EPILOGOS_PAIR_BATCH_ROWS")

    def add_staged(self, XA, NA, XB, NB, row0):
        """One part's two groups, resident.  row0 keys the null shuffle of the part's first row: the driver passes
        (file ordinal << 40) + row in the file -- known the moment a file is parsed, whatever the partition.  The part joins
        the pending batch; the batch is launched when it holds BATCH_ROWS rows (half a genome with the fused kernel, an eighth
        with the two-kernel path: a streaming run still counts while it parses) or 32 parts, or when anything needs its
        histograms."""
        S = self.S
        # widths of the widest part seen: an empty file pair (no columns) must not be the one that is remembered
        self.NA, self.NB = max(getattr(self, "NA", 0) or 0, NA or 0), max(getattr(self, "NB", 0) or 0, NB or 0)
        if XA.shape[0] == 0 or not NA or not NB:         # nothing to count (and the ABI rejects a zero width)
            self.parts.append((XA, XB, self.torch.empty((0, S), dtype=self.torch.int16, device=self.device),
                               self.torch.empty((0, S), dtype=self.torch.int16, device=self.device), row0, None))
            return len(self.parts) - 1
        self.parts.append((XA, XB, None, None, row0, None))
        pid = len(self.parts) - 1
        self._pending.append((pid, NA, NB))
        self._pending_rows += XA.shape[0]
        limit = int(self.BATCH_ROWS) if self.BATCH_ROWS else (8_000_000 if self.groupSize == -1 else 2_000_000)
        if self._pending_rows >= limit or len(self._pending) >= 32:
            self._flush()
        return pid

    def _flush(self):
        """Count pass of the pending batch and its null groups.  The default group sizes, a state model and widths the fused
        kernel takes: ONE launch does both (epg_pair_count_null_parts: a wave counts a tile of both groups, then draws its null
        groups -- the memory pipe and the VALU of a CU are busy at the same time without a second kernel).  Otherwise one launch
        for the count pass (epg_bin_hist_parts), then one for the null groups on the second stream (_start_null)."""
        if not self._pending:
            return
        eng, S = self.eng, self.S
        batch, self._pending, self._pending_rows = self._pending, [], 0
        XAs = [self.parts[pid][0] for pid, _na, _nb in batch]
        XBs = [self.parts[pid][1] for pid, _na, _nb in batch]
        k = len(batch)
        counts = self._acc(S) if self.sal == 1 else None  # counts over [A|B] = counts of A + counts of B (helpers.py:173)
        fused = None
        uniform = all(na == self.NA and nb == self.NB for _p, na, nb in batch)
        if (self.groupSize == -1 and uniform and all(self.parts[pid][4] is not None for pid, _na, _nb in batch)
                and os.environ.get("EPILOGOS_PAIR_FUSED", "1") != "0"):
            try:
                fused = eng.pair_count_null_parts(XAs, XBs, self.NA, self.NB, S, self.seed, [self.parts[pid][4] for pid, _na, _nb in batch],
                                                  counts=counts)
            except eng.EpilogosHipError as e:
                if e.code != -2:
                    raise
        if fused is not None:
            HAs, HBs, OAs, OBs = fused
            for i, (pid, _na, _nb) in enumerate(batch):
                XA, XB, _ha, _hb, row0, _null = self.parts[pid]
                self.parts[pid] = (XA, XB, HAs[i], HBs[i], row0, (OAs[i], OBs[i], None))
        else:
            widths = [na for _p, na, _nb in batch] + [nb for _p, _na, nb in batch]
            Hs, _ = eng.bin_hist_parts(XAs + XBs, widths, S, counts=counts)
            HAs, HBs = Hs[:k], Hs[k:]
            for i, (pid, _na, _nb) in enumerate(batch):
                XA, XB, _ha, _hb, row0, null = self.parts[pid]
                self.parts[pid] = (XA, XB, HAs[i], HBs[i], row0, null)
        if self.sal != 1:
            for i in range(k):
                eng.hist_s2_from_binhist_pair(HAs[i], HBs[i], S, counts=self._acc(S * S))
        if fused is None:
            self._start_null([pid for pid, _na, _nb in batch if self.parts[pid][4] is not None])

    def _start_null(self, pids):
        """The null groups' histograms of the parts `pids` (multivariate hypergeometric, from the real groups' histograms; they
        do not depend on exp_freq): one launch on the session's second stream, behind the launch that produced HA / HB."""
        if not pids:
            return
        t, eng = self.torch, self.eng
        NA, NB = self.NA, self.NB
        ga, gb = (NA, NB) if self.groupSize == -1 else (self.groupSize, self.groupSize)
        HAs, HBs = [self.parts[p][2] for p in pids], [self.parts[p][3] for p in pids]
        row0s = [self.parts[p][4] for p in pids]
        overlap = self.null_stream is not None and os.environ.get("EPILOGOS_NULL_OVERLAP", "1") != "0"
        done = None
        if overlap:
            # outputs from the main stream's pool (a second stream has a pool of its own in torch's allocator: every new session
            # would start with device mallocs), launch on the second stream between two events, no stream switch on the host
            main = t.cuda.current_stream()
            ready = t.cuda.Event()
            ready.record(main)
            self.null_stream.wait_event(ready)
            HnAs, HnBs = eng.null_hist_from_binhist_parts(HAs, HBs, NA + NB, self.S, ga, gb, self.seed, row0s, stream=self.null_stream)
            done = t.cuda.Event()
            done.record(self.null_stream)
            for x in (HAs[0], HBs[0], HnAs[0], HnBs[0]):
                x.record_stream(self.null_stream)        # used there: their memory must not be reused before it is through
        else:
            HnAs, HnBs = eng.null_hist_from_binhist_parts(HAs, HBs, NA + NB, self.S, ga, gb, self.seed, row0s)
        for i, p in enumerate(pids):
            XA, XB, HA, HB, row0, _ = self.parts[p]
            self.parts[p] = (XA, XB, HA, HB, row0, (HnAs[i], HnBs[i], done))

    def _null_of(self, pid):
        """(HnA, HnB) of part `pid`, ready for the current stream."""
        self._flush()
        if self.parts[pid][5] is None:
            self._start_null([pid])
        HnA, HnB, done = self.parts[pid][5]
        if done is not None:
            self.torch.cuda.current_stream().wait_event(done)
        return HnA, HnB

    # ---- multi-rank hand-over: the score pass of paired mode reads the two groups' histograms only
    n_export = 2

    def slice_part(self, pid, lo, hi, row0=None):
        self._flush()
        _XA, _XB, HA, HB, _, null = self.parts[pid]
        self.parts.append((None, None, _aligned_rows(HA, lo, hi), _aligned_rows(HB, lo, hi), row0, None))
        new = len(self.parts) - 1
        if null is not None:
            # the shuffle is keyed by (file, row in file): the rows' null groups are the ones drawn for the whole file
            HnA, HnB = self._null_of(pid)
            self.parts[new] = self.parts[new][:5] + ((_aligned_rows(HnA, lo, hi), _aligned_rows(HnB, lo, hi), None),)
        return new

    def export_rows(self, pid, lo, hi):
        self._flush()
        _XA, _XB, HA, HB, _, _null = self.parts[pid]
        return [HA[lo:hi], HB[lo:hi]]

    def import_rows(self, tensors, widths, row0=None):
        self.NA, self.NB = widths
        self.parts.append((None, None, tensors[0].to(self.device), tensors[1].to(self.device), row0, None))
        pid = len(self.parts) - 1
        if row0 is not None and self.parts[pid][2].shape[0]:
            self._start_null([pid])
        return pid

    def drop_part(self, pid):
        self._flush()
        self.parts[pid] = None

    def ensure_acc(self, N):
        self._flush()
        self._acc(self.S if self.sal == 1 else self.S * self.S)

    def all_reduce(self, d):
        self._flush()
        super().all_reduce(d)

    def finish_device(self, total_rows, N):
        self._flush()
        return super().finish_device(total_rows, N)

    def finish(self, total_rows, N):
        S = self.S
        return self._finish(total_rows, N, (S,) if self.sal == 1 else (S, S))

    def results_device(self, pid, keep=False):
        """Scores of A, B and the two null groups, deltas, null distances, STEP 4's per-bin reduction and the quiescence mask
        of part `pid` from its resident histograms, as device tensors."""
        self._flush()
        eng, S, NA, NB, q = self.eng, self.S, self.NA, self.NB, self.q
        XA, XB, HA, HB, row0, _null = self.parts[pid]
        if HA.shape[0] and row0 is None:
            raise ValueError("paired part %d has no shuffle key (row0)" % pid)
        HnA, HnB = self._null_of(pid) if HA.shape[0] else (None, None)
        if not keep:
            self.parts[pid] = None
        if HA.shape[0] == 0:
            t, dv = self.torch, self.device
            return {"delta": t.empty((0, S), dtype=t.float32, device=dv), "null": t.empty(0, dtype=t.float32, device=dv),
                    "quies": t.empty(0, dtype=t.uint8, device=dv), "rdist": t.empty(0, dtype=t.float32, device=dv),
                    "mdiff": t.empty(0, dtype=t.int32, device=dv)}
        ga, gb = (NA, NB) if self.groupSize == -1 else (self.groupSize, self.groupSize)
        # (the null groups' histograms were drawn straight from the real groups' when the part was counted: _start_null)
        if self.sal == 1:
            # one pass over the four histograms: table gathers, deltas, null distances and STEP 4's reduction (round 3); groups
            # too wide for the tables to sit in LDS take the separate passes below
            tabs = [self._s1_table(n) for n in (NA, NB, ga, gb)]
            try:
                delta, null, rdist, mdiff = eng.pair_scores_s1_from_binhist(HA, HB, HnA, HnB, S, NA, NB, ga, gb, *tabs)
                quies = eng.quiescent_from_binhist(HA, NA, HB, NB, S, self.qstate)
                return {"delta": delta, "null": null, "quies": quies, "rdist": rdist, "mdiff": mdiff}
            except self.eng.EpilogosHipError as e:
                if e.code != -2:
                    raise
            sA, sB = self._score_s1(HA, NA), self._score_s1(HB, NB)
            nA, nB = self._score_s1(HnA, ga), self._score_s1(HnB, gb)
        else:                                            # quirk Q9: null halves keep the original groups' permutation counts
            sA, _ = eng.score_s2_from_binhist(HA, NA, S, q, perms=NA * (NA - 1))
            sB, _ = eng.score_s2_from_binhist(HB, NB, S, q, perms=NB * (NB - 1))
            nA, _ = eng.score_s2_from_binhist(HnA, max(ga, NA), S, q, perms=NA * (NA - 1))
            nB, _ = eng.score_s2_from_binhist(HnB, max(gb, NB), S, q, perms=NB * (NB - 1))
        delta, _ = eng.pair_finish(sA, sB, want_dist=False)
        _, null = eng.pair_finish(nA, nB)
        rdist, mdiff = eng.pair_metrics(delta, roundtrip=True)     # what STEP 4 would recompute from the text
        quies = eng.quiescent_from_binhist(HA, NA, HB, NB, S, self.qstate)
        return {"delta": delta, "null": null, "quies": quies, "rdist": rdist, "mdiff": mdiff}

    def results_device_all(self, pids, keep=False):
        """results_device of several parts at once.  Paired S1 (the tables fit a CU's LDS): ONE launch of the fused pass over all the
        parts' histograms, quiescence masks included (epg_pair_scores_s1_parts) -- a launch per chromosome file paid the copy of
        the tables into LDS, the ramp and the tail 24 times (1.5 against 1.0 ms per 15 M bins).  Otherwise part by part."""
        self._flush()
        eng, S, NA, NB = self.eng, self.S, getattr(self, "NA", None), getattr(self, "NB", None)
        out = {}
        live = [pid for pid in pids if self.parts[pid] is not None and self.parts[pid][2].shape[0]]
        if self.sal == 1 and live and NA and NB:
            ga, gb = (NA, NB) if self.groupSize == -1 else (self.groupSize, self.groupSize)
            tabs = [self._s1_table(n) for n in (NA, NB, ga, gb)]
            quads = []
            for pid in live:
                if self.parts[pid][4] is None:
                    raise ValueError("paired part %d has no shuffle key (row0)" % pid)
                HnA, HnB = self._null_of(pid)
                quads.append((self.parts[pid][2], self.parts[pid][3], HnA, HnB))
            try:
                res = eng.pair_scores_s1_parts(quads, S, NA, NB, ga, gb, *tabs, qstate=self.qstate)
                for pid, r in zip(live, res):
                    out[pid] = r
                    if not keep:
                        self.parts[pid] = None
            except eng.EpilogosHipError as e:
                if e.code != -2:
                    raise
        for pid in pids:
            if pid not in out:
                out[pid] = self.results_device(pid, keep=keep)
        return [out[pid] for pid in pids]

    def _todo(self):
        self._flush()
        return [k for k, part in enumerate(self.parts) if part is not None and len(part) == 6 and (part[4] is not None or not part[2].shape[0])]

    def _begin(self, pids):
        todo = [pid for pid in pids if pid in set(self._todo())]
        self._early = dict(zip(todo, self.results_device_all(todo, keep=True)))

    def _settle(self):
        for pid in self._early:
            self.parts[pid] = None
        if self._early:
            self._ready = dict(self._early) if self._ready is None else dict(self._ready, **self._early)
        self._early = {}

    def _s1_widths(self):
        NA, NB = getattr(self, "NA", None), getattr(self, "NB", None)
        return [NA, NB] + ([self.groupSize] if self.groupSize != -1 else [])

    def results(self, pid):
        """Host arrays of part `pid`.  The first call computes the results of EVERY part still held (one launch, see
        results_device_all); the parts are then downloaded one by one as the driver asks for them."""
        self.check()
        if self._ready is None:
            todo = self._todo()
            self._ready = dict(zip(todo, self.results_device_all(todo)))
        r = self._ready.pop(pid) if pid in self._ready else self.results_device(pid)
        return {"delta": r["delta"].cpu().numpy(), "null": r["null"].cpu().numpy(), "quies": r["quies"].cpu().numpy().astype(bool),
                "rdist": r["rdist"].cpu().numpy(), "mdiff": r["mdiff"].cpu().numpy()}


def get():
    if _override is not None:
        return _override
    return HipBackend()


def set_for_testing(obj):
    """tests/ only."""
    global _override
    _override = obj

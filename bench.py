#!/usr/bin/env python3
"""
bench.py -- S1 scoring throughput of the MI355X engine on the BASELINE.json workload.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one whole S1 job over this rank's shard of bins, inputs resident in HBM (three launches + the collective):
    K1 per-bin histogram + state counts  ->  (N > 1: one RCCL all-reduce of the int64[18] count vector)
    -> [normalise to exp_freq (float32) + S1 score table] in one single-block kernel -> score pass -> float32 [bins, 18].
Workload: `--bins` (15,000,000 = whole-genome scale) synthetic bins x 833 biosamples x 18 states.
  --scaling strong (default; what BASELINE.json's north_star quotes): the `--bins` matrix is ONE genome,
      split over the ranks by the reference's splitRows rule (helpers.py:116-118): rank g holds bins
      [g*R//G, (g+1)*R//G); value = R * steps / max-over-ranks time.
  --scaling weak: every rank holds `--bins` bins (a genome-sized shard per GPU), the global matrix is G of them.
Synthetic states are i.i.d. with the empirical chr1 state frequencies, generated on device per fixed
global chunk seed, so the matrix does not depend on the GPU count.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel
(k_bin_hist, HBM-bound, 833 algorithmic bytes per bin) and `cpu_baseline` (the per-bin numpy loop of
oracle/rowloop_baseline.py on the host cores, N = 1 only).

The same line carries `configs`: BASELINE.json's configurations 3, 4 and 5 measured in the same run, device-resident,
through the sessions the command line uses (backend.HipBackend.open_single / open_paired -- the product's own sequence
of launches), each with its own roofline figure:
    "s2"     whole S2 job on the S1 matrix (K1, pair counts from the histograms, all-reduce, normalise, score pass),
    "s3"     S3 expected pass (matrix-core contraction) and score pass (LDS gathers) on `--s3-bins` bins,
    "paired" paired S1 on 379 + 342 biosamples: two count passes, all-reduce, the hypergeometric null groups, then scores of
             the four groups, deltas, null distances and STEP 4's per-bin reduction in one pass, and the quiescence mask.
    "s3" is the whole 15 M-bin genome (one repetition), "s3_small" the 2 M-bin job of earlier rounds.
`--configs none` skips them.  The headline step is the product's: a fresh backend._HipSingleSession per job, add_device(X, N) with
the default the command line's add_part runs under -- the histogram cache of a resident matrix of a GiB or more goes into another
memory class than the matrix (engine.alloc_hist, DESIGN.md 3: a bounded one-off walk per process, `placement.report`,
`placement.search_ms`), smaller matrices (the command line's chromosome parts) get plain allocations; the same jobs with a plain
allocation (add_device(place=False)) are timed after the timed region and reported as `placement.unplaced` -- the command-line
equivalent figure for a genome held as 24 parts.  `projected_speedup_8`: genome step / (step of an eighth + a measured one-rank RCCL
all-reduce), the prediction the first real 8-GPU run is to be held against.  After the headline: the same step as ONE hipGraph replay
(`graph_ms_per_step`), an RCCL self-test when there is a process group (`rccl_selftest`), the distribution variants of SURVEY.md
8d (`dist_variants`).  A secondary measurement that hangs ends the run with exit code 3 AFTER the line has been printed.
"""
import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FREQS = np.array([.00570, .00293, .00430, .00212, .03260, .10464, .00154, .00057, .01001, .00416, .01554, .00618,
                  .02498, .00262, .00140, .01412, .05563, .71097])
CHUNK_BINS = 1 << 20
SUB_BINS = 1 << 17
HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def host_cores():
    """(cores this process may use, how that was determined): the scheduler affinity capped by the cgroup CPU quota
    (the GPU boxes expose 256 hardware threads to a container that may run 16 of them at a time)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:
        pass
    if quota is not None and quota < aff:
        return quota, "cgroup cpu.max quota %d of %d schedulable hardware threads" % (quota, aff)
    return aff, "%d schedulable hardware threads (no smaller cgroup quota)" % aff


def cpu_baseline(n_biosamples, n_states, target_seconds=10.0):
    """Per-bin numpy loop (the reference's `-l` S1 loop shape) on the host: first ONE worker (the per-core rate the
    calibration in DESIGN.md / BASELINE.md quotes), then as many forked workers as the process may actually run at once.
    Bounded samples, before the process touches the GPU (it forks)."""
    from oracle import rowloop_baseline as rb
    from oracle import oracle_np as onp
    cores, how = host_cores()
    rng = np.random.default_rng(1234)
    p = FREQS[:n_states] / FREQS[:n_states].sum()
    sample = max(cores * 2000, 50_000)                       # distinct bins held in RAM (int8), >= 2000 per worker
    x = rng.choice(n_states, size=(sample, n_biosamples), p=p).astype(np.int8)
    q = onp.normalise(onp.expected_s1(x[:2000], n_states))
    bps1, secs1, _, bins1 = rb.timed_pool_run(x, q, n_states, 1, seconds=min(4.0, target_seconds))
    bps, secs, cores, bins = rb.timed_pool_run(x, q, n_states, cores, seconds=target_seconds)
    # second, fairer line: the vectorised numpy restatement (whole-array histogram + masked KL), one core
    xv = x[:20000]
    t0, nv = time.perf_counter(), 0
    while time.perf_counter() - t0 < 2.0:
        onp.score_s1(xv, onp.normalise(onp.expected_s1(xv, n_states)), n_states)
        nv += xv.shape[0]
    vec = nv / (time.perf_counter() - t0)
    return {"value": round(bps / 1e6, 6), "unit": "Mbins/s", "cores": cores, "kind": "port",
            "one_worker_bins_per_s": round(bps1, 1), "per_core_bins_per_s": round(bps / cores, 1),
            "vectorised_numpy_one_core_bins_per_s": round(vec, 1),
            "parallel_efficiency": round(bps / (cores * bps1), 3),
            "sample": "%d bin-scorings in %.1f s wall (plus %d in %.1f s on one worker) over %d distinct synthetic bins x %d "
                      "biosamples (same state frequencies as the GPU workload): per-bin numpy loop of oracle/rowloop_baseline.py "
                      "(np.unique + numpy.ma p*log2(p/q), the reference's -l S1 loop shape, scores.py:309-344,539-550) on %d "
                      "forked workers = %s" % (bins, secs, bins1, secs1, sample, n_biosamples, cores, how)}


def k1_source_sha():
    """Hash of the sources k_bin_hist is built from: profiles/hbm_traffic.json records the hash its PMC numbers were
    taken with, and a number measured on other kernel code is not reported."""
    import hashlib
    h = hashlib.sha256()
    for name in ("epg_count.h", "epg_common.h"):
        h.update((ROOT / "epilogos_amd" / "csrc" / name).read_bytes())
    src = (ROOT / "epilogos_amd" / "csrc" / "epg_s1.hip").read_text()
    h.update(src[:src.index("// Any S <= 127")].encode())    # store_staged + k_bin_hist
    return h.hexdigest()[:16]


def generate_shard(torch, X, n_biosamples, n_states, bin0, dist="chr1", seed=1234):
    """Fill X[:, :N] with synthetic states for global bins [bin0, bin0 + R): global chunk k of 2^20 bins is drawn
    from torch.Generator seeded 1234 + k in fixed sub-blocks, so the matrix does not depend on the GPU count.
    dist: "chr1" = i.i.d. empirical chr1 state frequencies (the headline workload); "uniform" = uniform
    states (contention-free control); "correlated" = each bin copies its predecessor with probability 0.83 and 41 % of
    bins are forced all-quiescent (the stress shape for shared-counter histograms)."""
    R = X.shape[0]
    dev = X.device
    p = FREQS[:n_states] / FREQS[:n_states].sum()
    if dist == "uniform":
        p = np.full(n_states, 1.0 / n_states)
    bounds = torch.tensor(np.cumsum(p)[:-1], dtype=torch.float32, device=dev)
    X.fill_(-1)
    gen = torch.Generator(device=dev)
    k0, k1 = bin0 // CHUNK_BINS, (bin0 + R - 1) // CHUNK_BINS
    for k in range(k0, k1 + 1):
        gen.manual_seed(seed + k)
        for sub in range(CHUNK_BINS // SUB_BINS):
            g0 = k * CHUNK_BINS + sub * SUB_BINS
            u = torch.rand((SUB_BINS, n_biosamples), generator=gen, device=dev, dtype=torch.float32)
            lo, hi = max(g0, bin0), min(g0 + SUB_BINS, bin0 + R)
            if lo >= hi:
                continue
            st = torch.bucketize(u[lo - g0:hi - g0], bounds, right=True).to(torch.int8)
            X[lo - bin0:hi - bin0, :n_biosamples] = st
            del u, st
    if dist == "correlated":
        gen.manual_seed(99 + bin0)
        step = 1 << 20
        for r0 in range(0, R, step):
            r1 = min(r0 + step, R)
            r = torch.rand(r1 - r0, generator=gen, device=dev)
            quiescent = r < 0.41
            X[r0:r1][quiescent, :n_biosamples] = n_states - 1
            copy = (r >= 0.41) & (r < 0.41 + 0.83 * 0.59)
            idx = torch.nonzero(copy, as_tuple=False).flatten() + r0
            idx = idx[idx > 0]
            X[idx, :n_biosamples] = X[idx - 1, :n_biosamples]    # copies the (already final) predecessor of most bins


LDS_GATHER_PEAK = 256 * 2.4e9 / 2 * 64   # ds_read_b32 lanes/s of the chip: one wave instruction per 2 cycles and CU (MI355X_MICROARCH.md)
FP4_DENSE_PEAK = 10.0e15                 # dense fp4 MFMA FLOP/s (MI355X_MICROARCH.md; 2x the fp8 figure)


class _StdoutToStderr:
    """RCCL prints a version banner to the C-level stdout when a communicator is created (flushed whenever the C buffer is, i.e.
    possibly AFTER this program's one JSON line).  One-rank groups are joined inside this context: file descriptor 1 points at
    stderr while the communicator comes up, and the C buffers are flushed before it is put back."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self.libc = ctypes.CDLL(None)
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def _max_over_ranks(torch, d, value, dev):
    if d.world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    d.dist.all_reduce(t, op=d.dist.ReduceOp.MAX)
    return float(t.item())


def _event_ms(torch, fn, reps=3):
    """Median device time of fn() between two events on the current stream (one untimed call first)."""
    ts = []
    for k in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if k:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def bench_single(torch, be, d, sal, X, N, S, R, R_global, world, reps, fence):
    """One whole single-group job of saliency 2 or 3 per repetition, through the session the command line uses
    (backend._HipSingleSession: add_device -> all_reduce -> launch [= finish_device + the score passes]; its finish() -- count
    check and exp_freq download, a host synchronisation -- after the clock), inputs resident.  Wall time
    between fences (max over ranks) and device time of the phases from events on the launch stream."""
    dev = X.device
    walls, phases = [], []
    for rep in range(reps + 1):                                   # the first repetition warms allocations and attributes up
        sess = be.open_single(S, sal)
        fence()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        e[0].record()
        pid = sess.add_device(X, N)                               # STEP 1 (expected counts; histograms / matrix stay resident)
        e[1].record()
        sess.ensure_acc(N)
        sess.all_reduce(d)                                        # the one collective: int64[S*S] or int32[N*N*S*S]
        sess.finish_device(R_global, N)                           # STEP 2 (S3 checks its counts at once; S2: in finish())
        e[2].record()
        sess.launch_scores([pid])                                 # STEP 3
        o32 = sess.early_scores(pid)
        e[3].record()
        fence()
        wall = time.perf_counter() - t0
        if rep:
            walls.append(_max_over_ranks(torch, d, wall, dev))
            phases.append([e[i].elapsed_time(e[i + 1]) for i in range(3)])
        sess.finish(R_global, N)                                  # the host side: count check, exp_freq download
        ok = bool(torch.isfinite(o32[:: max(R // 4096, 1)]).all()) if R else True
        del sess, o32
    wall_ms = float(np.median(walls)) * 1e3
    exp_ms, comb_ms, score_ms = (float(v) for v in np.median(np.array(phases), axis=0))
    out = {"bins_total": R_global, "bins_per_gpu": R, "biosamples": N, "states": S, "saliency": sal, "reps": reps,
           "job_ms": round(wall_ms, 3), "value": round(R_global / wall_ms / 1e3, 3), "unit": "Mbins/s", "scores_finite": ok,
           "phases_ms": {"expected": round(exp_ms, 3), "allreduce+normalise": round(comb_ms, 3), "scores": round(score_ms, 3)},
           "path": "backend._HipSingleSession (the command line's session), device-resident"}
    if sal == 2:
        # bytes the job has to move per bin: X read, H written (the pair counts come out of the count pass's own launch since
        # round 5: no pass of their own over H), H read + float32 scores written
        bpb = N + 2 * S + 2 * S + 4 * S
        eng = be.engine
        H = eng.alloc_hist(X, N, S)                                         # (where the session's jobs had it)
        c2 = torch.zeros(S * S, dtype=torch.int64, device=dev)
        eng.bin_hist_s2(X, N, S, counts2=c2, H=H)
        q2 = eng.normalise(c2)
        o = torch.empty((R, S), dtype=torch.float32, device=dev)
        ws = eng.workspace(2, 0, N, S, device=dev)
        k1 = _event_ms(torch, lambda: eng.bin_hist_s2(X, N, S, counts2=c2, H=H))
        k0 = _event_ms(torch, lambda: eng.bin_hist(X, N, S, want_counts=False, H=H))
        kc = _event_ms(torch, lambda: eng.hist_s2_from_binhist(H, S, counts=c2))
        ks = _event_ms(torch, lambda: eng.score_s2_from_binhist(H, N, S, q2, out32=o, ws=ws))
        out["kernels_ms"] = {"k_bin_hist_s2 (count pass + pair counts, one launch)": round(k1, 4), "k_score_s2_bin(+tables)": round(ks, 4),
                             "for comparison: k_bin_hist alone": round(k0, 4), "for comparison: k_s2_hist_wave (the separate pair-count pass of rounds 2-4)": round(kc, 4)}
        gb = R * bpb / (exp_ms + comb_ms + score_ms) / 1e6
        out["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "algorithmic_bytes_per_bin": bpb,
                           "achieved": round(gb, 1), "frac": round(gb / HBM_PEAK_GBPS, 4),
                           "what": "whole job (device time of the three phases) against the bytes it has to move",
                           "kernel_fracs": {"k_bin_hist_s2 (N + 2S B/bin)": round(R * (N + 2 * S) / k1 / 1e6 / HBM_PEAK_GBPS, 4),
                                            "k_score_s2_bin (6S B/bin)": round(R * 6 * S / ks / 1e6 / HBM_PEAK_GBPS, 4)}}
        del H
    else:
        pairs = float(R) * N * (N - 1)
        M = N * (S - 1) if R >= 262144 else N * S                 # the reduced contraction leaves one state per biosample out
        flops = float(R) * M * M                                  # 2 * R * M^2 / 2: the upper triangle of E^T E
        out["ms_per_Mbins"] = {"expected": round(exp_ms / R * 1e6, 3), "scores": round(score_ms / R * 1e6, 3)}
        out["pair_terms_per_s"] = {"expected": float("%.4g" % (pairs / exp_ms * 1e3)), "scores": float("%.4g" % (pairs / score_ms * 1e3))}
        out["roofline"] = {"expected": {"bound": "mfma", "unit": "TFLOP/s", "peak": FP4_DENSE_PEAK / 1e12,
                                        "achieved": round(flops / exp_ms / 1e9, 1), "frac": round(flops / exp_ms * 1e3 / FP4_DENSE_PEAK, 4),
                                        "flops": "R * M^2 with M = N * (S - 1) rows of the reduced one-hot operand (upper triangle, "
                                                 "fp4 E2M1); the phase also holds the transpose, the operand build and the reconstruction"},
                           "scores": {"bound": "lds", "unit": "gathers/s", "peak": LDS_GATHER_PEAK, "achieved": float("%.4g" % (pairs / score_ms * 1e3)),
                                      "frac": round(pairs / score_ms * 1e3 / LDS_GATHER_PEAK, 4),
                                      "what": "N (N - 1) table terms per bin against the chip's ds_read_b32 lane rate; the phase "
                                              "also holds the table build and the transpose"}}
        # the kernel's memory side (the 990 MB fixed-point table is re-streamed once per group of workgroups that walk it together):
        # FETCH_SIZE of the PMC pass of tools/profile_bench.sh on this kernel source and shape, when there is one
        try:
            rec = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())
            fb = rec.get("k_s3_score_bl_fetch_bytes_per_launch_%d_%d" % (R, N)) if rec.get("k1_source_sha") == k1_source_sha() else None
        except Exception:
            fb = None
        if fb:
            gbps = fb / score_ms / 1e6
            out["roofline"]["scores"]["memory_side"] = {
                "fetch_bytes_per_bin": round(fb / R), "fetch_GBps_over_the_score_phase": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4),
                "table_passes": round(fb / (990e6 if N == 833 and S == 18 else max(fb, 1)), 1) if N == 833 and S == 18 else None,
                "frac_of_copy_rate": round(gbps / 6290.0, 4),
                "what": "FETCH_SIZE x 1024 x 2 of k_s3_score_bl's launch (PMC pass, profiles/hbm_traffic.json) over the phase's device time, "
                        "against the 8 TB/s spec and the 6.29 TB/s a copy reaches: as large a fraction of its ceiling as the gathers are of "
                        "theirs.  Which one the kernel waits for was measured (profiles/r06f_s3_score_launch_groups.txt): 35 % less fetch "
                        "(one launch per 1 M bins) made it 1.3 % slower -- the gathers bind, the table stream is next"}
    return out


# hg19 chromosome lengths (chr1..chr22, X, Y; base pairs -- 200-bp bins are length // 200): the shape of the parts the command
# line feeds a session with, one file per chromosome
HG19_BP = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431, 135534747, 135006516,
           133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983, 63025520, 48129895, 51304566,
           155270560, 59373566]


def chromosome_parts(R_global, lo, hi):
    """The genome of R_global bins cut into 24 files in hg19's proportions; -> [(file ordinal, first row in the file, global
    first bin, global end bin)] of the pieces that fall into this rank's bin range [lo, hi) -- what driver.plan_partition gives
    a rank of the command line."""
    w = np.array(HG19_BP, dtype=np.float64)
    edges = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * R_global)]).astype(np.int64)
    edges[-1] = R_global
    parts = []
    for f in range(len(HG19_BP)):
        a, b = max(lo, int(edges[f])), min(hi, int(edges[f + 1]))
        if a < b:
            parts.append((f, a - int(edges[f]), a, b))
    return parts


def bench_paired(torch, be, d, R, R_global, bin0, S, world, reps, fence, dev, dist_name, NA=379, NB=342):
    """Paired S1 (BASELINE config 5: male vs female, 379 + 342 biosamples) through backend._HipPairedSession: both groups'
    count passes, the all-reduce of the [A|B] counts, normalise, the hypergeometric null groups, four score passes, deltas, null
    distances, STEP 4's per-bin reduction and the quiescence mask -- everything the command line computes before it writes.
    The session is fed the way the command line feeds it: one part per chromosome file (24 in hg19's proportions, natural
    order), each keyed by (file, row); the session counts the parts in batches (one kernel per batch counts both groups and
    draws the null groups)."""
    from epilogos_amd.driver import shuffle_key
    eng = be.engine
    XA, XB = eng.alloc_states(R, NA, device=dev), eng.alloc_states(R, NB, device=dev)
    generate_shard(torch, XA, NA, S, bin0, dist=dist_name, seed=4321)
    generate_shard(torch, XB, NB, S, bin0, dist=dist_name, seed=8765)
    parts = chromosome_parts(R_global, bin0, bin0 + R)
    walls, phases = [], []
    for rep in range(reps + 1):
        sess = be.open_paired(S, 1, S - 1, -1, 20240229)
        fence()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter()
        e[0].record()
        pids = [sess.add_staged(XA[a - bin0:b - bin0], NA, XB[a - bin0:b - bin0], NB, shuffle_key(f, r0)) for f, r0, a, b in parts]
        e[1].record()
        host_count_ms = (time.perf_counter() - t0) * 1e3
        sess.ensure_acc(NA + NB)
        sess.all_reduce(d)
        sess.finish_device(R_global, NA + NB)
        e[2].record()
        sess.launch_scores(pids)                               # one launch over all the parts
        res = [sess._early[p] for p in pids]
        e[3].record()
        fence()
        wall = time.perf_counter() - t0
        if rep:
            walls.append(_max_over_ranks(torch, d, wall, dev))
            phases.append([e[i].elapsed_time(e[i + 1]) for i in range(3)])
        sess.finish(R_global, NA + NB)                         # host side: count check, table verification
        ok = all(bool(torch.isfinite(r["delta"][:: max(r["delta"].shape[0] // 512, 1)]).all()) and
                 bool(torch.isfinite(r["null"][:: max(r["null"].shape[0] // 512, 1)]).all()) for r in res)
        nq = int(sum(int(r["quies"].sum().item()) for r in res))
        patched = sess.tables_patched
        del sess, res
    wall_ms = float(np.median(walls)) * 1e3
    exp_ms, comb_ms, res_ms = (float(v) for v in np.median(np.array(phases), axis=0))
    # bytes per bin: both matrices read + two histograms written; null groups: 2 H read, 2 written; the fused score pass: 4 H read,
    # delta (4 S) + null distance + STEP 4's distance and state (12) written; quiescence: 2 H read, 1 B written
    bpb = (NA + NB) + 2 * 2 * S + 4 * 2 * S + (4 * 2 * S + 4 * S + 12) + (2 * 2 * S + 1)
    gb = R * bpb / (exp_ms + comb_ms + res_ms) / 1e6
    return {"bins_total": R_global, "bins_per_gpu": R, "biosamples": [NA, NB], "states": S, "saliency": 1, "reps": reps,
            "parts": len(parts), "parts_what": "one per chromosome file in hg19's proportions, keyed (file, row) for the null shuffle",
            "job_ms": round(wall_ms, 3), "value": round(R_global / wall_ms / 1e3, 3), "unit": "Mbins/s",
            "outputs_finite": ok, "quiescent_bins": nq, "s1_tables_equal_numpy_reference": patched == 0,
            "host_enqueue_ms_of_the_count_phase": round(host_count_ms, 3),
            "phases_ms": {"count pass + null draw of the batches that filled up while the %d parts were added (default group sizes: one "
                          "kernel per batch, k_pair_count_null; otherwise count launch + sampler launch on the second stream)" % len(parts): round(exp_ms, 3),
                          "count pass + null draw of the last batch, all-reduce, normalise, tables": round(comb_ms, 3),
                          "scores/deltas/null distances/metrics/quiescence of all parts (one launch)": round(res_ms, 3)},
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "algorithmic_bytes_per_bin": bpb,
                         "achieved": round(gb, 1), "frac": round(gb / HBM_PEAK_GBPS, 4),
                         "what": "whole job (device time of the three phases on the main stream) against the bytes it has to move; the "
                                 "count + null-draw kernel is bound by its VALU instruction count (Philox + selection sampling on top of the "
                                 "counting core: 110 wave instructions per bin), see DESIGN.md 3"},
            "path": "backend._HipPairedSession (the command line's session), device-resident"}


def rank_identity(torch, dev, local_rank):
    """Which GPU this rank really drives: HIP device index and name, PCI address, the NUMA node of that PCI device, the
    *_VISIBLE_DEVICES variables, the cores the process may run on (the first 8-GPU run should explain itself)."""
    who = {"local_rank": local_rank, "hip_device": dev.index, "pid": os.getpid(), "cores_allowed": len(os.sched_getaffinity(0))}
    try:
        pr = torch.cuda.get_device_properties(dev)
        who["name"] = pr.name
        who["gcn_arch"] = getattr(pr, "gcnArchName", None)
        who["cus"] = pr.multi_processor_count
        who["hbm_GiB"] = round(pr.total_memory / 2**30, 1)
        dom, bus, devid = (getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if bus is not None:
            addr = "%04x:%02x:%02x.0" % (dom or 0, bus, devid or 0)
            who["pci"] = addr
            node = Path("/sys/bus/pci/devices") / addr / "numa_node"
            who["numa_node"] = int(node.read_text()) if node.exists() else None
    except Exception as e:
        who["error"] = repr(e)[:120]
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        if k in os.environ:
            who[k] = os.environ[k]
    return who


def rccl_selftest(torch, dist, dev, rank, world, N, S):
    """Every collective the multi-rank command line uses, once, checked against numbers: the all-reduce of the S1 count vector
    (int64[S]) and of the S3 count tensor at its real size (int32 [N, N, S, S], 899 MB at N = 833), the device-to-device
    hand-over of border pieces (driver._redistribute: send/recv, here a ring), the broadcast of the null seed
    (driver: int64[1]) and the all_gather bench.py itself uses.  -> {leg: {"ok", "ms"}}; a leg that raises is reported, not fatal."""
    out = {"world": world, "backend": dist.get_backend()}
    rccl = dist.get_backend() == "nccl"
    full = N * N * S * S
    if not rccl:                       # a host-side backend (the CPU / one-GPU tests): tensors on the host, the big one 1/64
        dev, full = torch.device("cpu"), max(full // 64, 1 << 20)
    sync = torch.cuda.synchronize if rccl else (lambda: None)

    def leg(name, fn):
        t0 = time.perf_counter()
        try:
            ok = bool(fn())
            sync()
            out[name] = {"ok": ok, "ms": round((time.perf_counter() - t0) * 1e3, 2)}
        except Exception as e:
            out[name] = {"ok": False, "error": repr(e)[:200]}

    def ar_small():
        t = torch.arange(S, dtype=torch.int64, device=dev) * (rank + 1)
        dist.all_reduce(t)
        return torch.equal(t.cpu(), torch.arange(S, dtype=torch.int64) * (world * (world + 1) // 2))

    def ar_big():
        n = full
        t = torch.full((n,), rank + 1, dtype=torch.int32, device=dev)
        t[::1000003] += 7 * rank
        sync()
        t0 = time.perf_counter()
        dist.all_reduce(t)
        sync()
        ms = (time.perf_counter() - t0) * 1e3
        out["allreduce_int32_%d_ms" % n] = round(ms, 3)
        out["allreduce_int32_busbw_GBps"] = round(2 * (world - 1) / world * n * 4 / ms / 1e6, 1) if world > 1 else None
        base = world * (world + 1) // 2
        want = torch.full((n,), base, dtype=torch.int32, device=dev)
        want[::1000003] += 7 * (world * (world - 1) // 2)
        return torch.equal(t, want)

    def ring():
        if world == 1:
            return True
        n = 16 << 20                                                        # a 64 MB border piece of int32
        mine = torch.full((n,), rank, dtype=torch.int32, device=dev)
        got = torch.empty_like(mine)
        nxt, prv = (rank + 1) % world, (rank - 1) % world
        ops = [dist.P2POp(dist.isend, mine, nxt), dist.P2POp(dist.irecv, got, prv)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return bool((got == prv).all().item())

    def plain_p2p():
        if world == 1:
            return True
        # the driver's hand-over is plain send / recv between pairs (lazy communicator set-up on first use): even ranks send first
        t = torch.full((1 << 20,), rank, dtype=torch.int16, device=dev)
        g = torch.empty_like(t)
        peer = rank ^ 1
        if peer >= world:
            return True
        if rank % 2 == 0:
            dist.send(t, peer)
            dist.recv(g, peer)
        else:
            dist.recv(g, peer)
            dist.send(t, peer)
        return bool((g == peer).all().item())

    def bcast():
        t = torch.tensor([20240229 if rank == 0 else -1], dtype=torch.int64, device=dev)
        dist.broadcast(t, 0)
        return int(t.item()) == 20240229

    def gather():
        mine = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        return [int(v.item()) for v in allv] == list(range(world))

    leg("all_reduce int64[%d]" % S, ar_small)
    leg("all_reduce int32[%d] (the S3 count tensor%s)" % (full, "" if rccl else ", 1/64 of it: host backend"), ar_big)
    leg("send/recv ring of 64 MB (batch_isend_irecv)", ring)
    leg("send/recv pairs (driver._redistribute's hand-over)", plain_p2p)
    leg("broadcast int64[1] (null seed)", bcast)
    leg("all_gather float64[1]", gather)
    out["ok"] = all(v.get("ok", True) for v in out.values() if isinstance(v, dict))
    return out


def launch_command(gpus, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)


_KEEPER_SRC = r"""
import signal, sys
for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
    signal.signal(sig, signal.SIG_IGN)
last = None
for l in sys.stdin.buffer:
    if l.endswith(b"\n"):
        last = l
if last is not None:
    sys.stdout.buffer.write(last)
    sys.stdout.flush()
"""


class LineKeeper:
    """Rank 0 of a run with more than one rank: the ONE JSON line is printed by a small child process (started before the first GPU
    call, GPU-free, a session of its own) that holds the latest version rank 0 has handed it and prints it when rank 0's end of the
    pipe closes -- because rank 0 finished, or because it died in a secondary measurement (an abort inside a collective, a SIGTERM
    from the launcher after another rank failed): no box of this pool has more than one GPU, so every multi-rank leg after the timed
    region runs for the first time on the driver's node, and none of them may take the headline with it."""

    def __init__(self):
        import subprocess
        self.p = subprocess.Popen([sys.executable, "-c", _KEEPER_SRC], stdin=subprocess.PIPE, start_new_session=True)

    def keep(self, text):
        try:
            self.p.stdin.write(text.encode() + b"\n")
            self.p.stdin.flush()
            return True
        except (OSError, ValueError):
            return False

    def final(self, text):
        """Hand the final line over and wait until it is out; False = the keeper is gone (the caller prints)."""
        ok = self.keep(text)
        try:
            self.p.stdin.close()
        except OSError:
            ok = False
        try:
            ok = self.p.wait(timeout=20) == 0 and ok
        except Exception:
            ok = False
        return ok


def launch_ranks(gpus, argv):
    """One process per GPU under torch.distributed.run, as a child of this (GPU-free) process; -> its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores()[0] // gpus)))
    cmd = launch_command(gpus, argv, port)
    if os.environ.get("EPILOGOS_LAUNCH_DRYRUN"):
        print(" ".join(cmd))
        return 0
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def allreduce_child(n):
    """`bench.py --allreduce-child N`: a process of its own joins a one-rank RCCL group and times the all-reduce of the count vector
    (int64[N]) back to back; one JSON line.  The parent runs it with a timeout AFTER its timed region: a communicator that does not
    come up (seen once on a box of the pool: > 300 s inside a process that had walked 96 GiB of device memory) costs the parent a
    missing number, not its line or its exit code -- and RCCL's version banner stays in the child's stdout."""
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
    t = torch.zeros(n, dtype=torch.int64, device=dev)
    for _ in range(20):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        dist.all_reduce(t)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    dist.destroy_process_group()
    sys.stderr.flush()
    print(json.dumps({"allreduce_us": round(us, 2)}), flush=True)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--allreduce-child":
        return allreduce_child(int(sys.argv[2]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bins", type=int, default=15_000_000, help="bins of the genome (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None,
                    help="default strong: ONE --bins genome split over the GPUs")
    ap.add_argument("--biosamples", type=int, default=833)
    ap.add_argument("--states", type=int, default=18)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--packed", action="store_true", help="row pitch = biosamples (unaligned rows) instead of 16-byte padded")
    ap.add_argument("--dist", choices=["chr1", "uniform", "correlated"], default="chr1", help="synthetic state distribution")
    ap.add_argument("--configs", default="s2,s3,s3_small,paired", help="BASELINE configs 3-5 measured next to the headline: any of s2,s3,s3_small,paired; none")
    ap.add_argument("--s3-bins", type=int, default=0, help="bins of the S3 measurement (whole job, all ranks); 0 = the genome (--bins), one repetition")
    ap.add_argument("--s3-small-bins", type=int, default=2_000_000, help="bins of the s3_small measurement (--config-reps repetitions)")
    ap.add_argument("--dist-variants", type=int, default=1, help="one GPU: K1 / job time on the uniform and the row-correlated matrix (SURVEY 8d); 0 = skip")
    ap.add_argument("--graph-leg", type=int, default=1, help="after the headline: the step as one hipGraph replay (graph_ms_per_step); 0 = skip")
    ap.add_argument("--config-reps", type=int, default=3)
    ap.add_argument("--extras-deadline", type=int, default=300,
                    help="seconds the measurements after the timed region (--configs, the unplaced jobs, the self-test, the graph leg) may take before rank 0 "
                         "prints the line without the rest and every rank exits; 0 = no limit")
    ap.add_argument("--graph", action="store_true", help="capture the step (K1, all-reduce, combine, score) in a hipGraph and replay it "
                                                         "(kernels_ms then comes from an un-captured probe after the timed region)")
    ap.add_argument("--path", choices=["session", "engine"], default="session",
                    help="what a headline step runs: session = backend._HipSingleSession, the command line's own sequence of calls "
                         "(add_device -> all_reduce -> finish_device -> scores_device; per-step allocations like the product's); "
                         "engine = the bare ABI calls on preallocated buffers (always used by --graph)")
    ap.add_argument("--shard-bins", type=int, default=1_875_000, help="one GPU: both paths are also timed on a shard of this "
                                                                        "many bins (an eighth of the genome: the 8-GPU share); 0 = skip")
    ap.add_argument("--allreduce-leg", type=int, default=1, help="one GPU without --pg: join a one-rank RCCL group AFTER the timed region to time the count vector's all-reduce (for projected_speedup_8); 0 = skip")
    ap.add_argument("--pg", action="store_true", help="one GPU: still join a one-rank RCCL process group, so that the step contains the all-reduce")
    ap.add_argument("--placement-experiment", type=int, default=1, help="0: skip the jobs with an unplaced histogram cache after the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to test the "
                                                      "multi-rank path on a box with one GPU, ranks then share cuda:0)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a plain start with --gpus N: this process becomes the launcher (the reference's one command submits its own
        # workers, run.py:190-279,454-505) -- it has not imported torch or touched a GPU, starts
        # `python -m torch.distributed.run ... bench.py <same arguments>` as a CHILD (never exec), lets rank 0's one JSON
        # line through on the inherited stdout and leaves with the child's exit code
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    real_stdout = os.dup(1)                                          # the line goes here, whatever happens to fd 1 later
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    scaling = args.scaling or "strong"              # one GPU: the two coincide
    N, S = args.biosamples, args.states
    if scaling == "strong":                  # the reference's splitRows rule on ONE genome (helpers.py:116-118)
        bin0, bin1 = rank * args.bins // world, (rank + 1) * args.bins // world
        R_global = args.bins
    else:
        bin0, bin1 = rank * args.bins, (rank + 1) * args.bins
        R_global = args.bins * world
    R = bin1 - bin0

    keeper = None
    if rank == 0 and world > 1:                                      # (before the first GPU call)
        try:
            keeper = LineKeeper()
        except OSError:
            keeper = None                                            # no child to be had: rank 0 prints the line itself
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(N, S)      # before any GPU initialisation: it forks
        except Exception as e:            # a host without fork / enough memory must not take the GPU measurement down
            cpu = {"value": None, "unit": "Mbins/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    import torch
    import torch.distributed as dist
    from epilogos_amd import engine
    engine.require_gpu()
    if args.backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
        if world > torch.cuda.device_count():
            os.environ["EPILOGOS_PLACEMENT"] = "0"       # ranks share a GPU (tests): no search that holds 4 GiB blocks per process
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_pg = world > 1 or args.pg
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        limit = datetime.timedelta(seconds=max(600, 2 * args.extras_deadline))     # a lost rank ends the run instead of hanging it
        if args.backend == "nccl" and world == 1:
            with _StdoutToStderr():                                   # (the banner of a one-rank group must not follow the JSON line)
                dist.init_process_group(backend="nccl", device_id=dev, timeout=limit)
                dist.all_reduce(torch.zeros(1, device=dev))
                torch.cuda.synchronize()
        elif args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend=args.backend, timeout=limit)

    # ---- resident inputs and preallocated outputs: plain allocations, like the sessions of the command line
    if args.packed:
        flat = torch.empty(R * N + 64, dtype=torch.int8, device=dev)
        X = flat[:R * N].view(R, N)
    else:
        X = engine.alloc_states(R, N, device=dev)
    generate_shard(torch, X, N, S, bin0, dist=args.dist)
    counts = torch.zeros(S, dtype=torch.int64, device=dev)
    q = torch.empty(S, dtype=torch.float32, device=dev)
    use_session = args.path == "session" and not args.graph
    # the bare-ABI path works on buffers of its own; the session path allocates per job (nothing is held for it here)
    out32 = None if use_session else torch.empty((R, S), dtype=torch.float32, device=dev)
    ws_s1 = engine.workspace(1, 0, N, S, device=dev)
    H = None if use_session else engine.alloc_hist(X, N, S)                # (the library's allocator for a resident matrix)
    last_counts = torch.zeros(S, dtype=torch.int64, device=dev)

    from epilogos_amd import backend as _backend
    from epilogos_amd.driver import _Dist
    be = _backend.HipBackend(device=dev)
    d = _Dist()
    last = {}

    def step_engine(Xs, Hs, outs, e=None, keep=False):
        """The bare ABI sequence on preallocated buffers."""
        if e is not None:
            e[0].record()
        engine.bin_hist(Xs, N, S, counts=counts, H=Hs)                     # STEP 1: expected pass (counts start zeroed)
        if e is not None:
            e[1].record()
        if use_pg:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)                  # the single collective (144 bytes)
        if e is not None:
            e[2].record()
        if keep:
            last_counts.copy_(counts)                                      # outside the timed steps: for the sanity check
        # STEP 2 + STEP 3: normalise + table in one kernel (it leaves counts zeroed for the next job), score pass
        engine.combine_score_s1(counts, Hs, N, S, q=q, out32=outs, ws=ws_s1, rezero=True)
        if e is not None:
            e[3].record()

    def step_session(Xs, e=None, place=None, finish=False):
        """One whole S1 job through the product's session: a fresh session per job; add_device with ITS default (place=None: the
        rule the command line's add_part runs under -- engine.alloc_hist places the histogram cache of a matrix of a GiB or more in
        the device's home block, found by the first job of the process; place=False is the comparison with a plain allocation),
        the count vector, the tables and the scores allocated by the session (torch's caching allocator hands the previous job's
        blocks back)."""
        last.clear()                                                       # (the previous job's session and scores are released first)
        sess = be.open_single(S, 1)
        if e is not None:
            e[0].record()
        pid = sess.add_device(Xs, N, place=place)                          # STEP 1: k_bin_hist (+ counts)
        if e is not None:
            e[1].record()
        sess.ensure_acc(N)
        sess.all_reduce(d)                                                 # the single collective (144 bytes)
        if e is not None:
            e[2].record()
        total = R_global if Xs.shape[0] == R else Xs.shape[0]              # (the one-GPU shard measurement is a job of its own)
        sess.launch(total, N, [pid])                                       # STEP 2 + the S1 table (k_s1_combine) + STEP 3, no host sync
        last["out"] = sess.early_scores(pid)
        if e is not None:
            e[3].record()
        if finish:
            sess.finish(total, N)                                       # count check + table verification + exp_freq download
        last["sess"] = sess

    def fence():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(Xs, Hs, outs, steps, warmup, graph=False, events=True, session=False, place=None, finish=False):
        """-> (wall seconds of `steps` steps, per-step event tuples or [], host enqueue times, this rank's own seconds:
        from the common start to the moment ITS last step was over, before the closing barrier)."""
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(steps)] if (events and not graph) else []
        one = (lambda e=None: step_session(Xs, e, place, finish)) if session else (lambda e=None: step_engine(Xs, Hs, outs, e))
        g = None
        if graph:
            # the whole step as ONE hipGraph launch: on a 1.9 M-bin shard the step is ~0.35 ms and the host side of three
            # launches + a collective (ctypes, torch.distributed) is no longer hidden behind it
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    one()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one()
        for _ in range(warmup):
            g.replay() if g is not None else one()
        fence()
        t0 = time.perf_counter()
        host_t = []
        for k in range(steps):
            if g is not None:
                g.replay()
            else:
                one(ev[k] if ev else None)
            host_t.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        fence()
        return time.perf_counter() - t0, ev, host_t, own

    if use_session:
        step_session(X)                                                    # untimed: sanity of one whole job
        # engine.alloc_hist searches at the 2nd and (if needed) the 4th job on a matrix: whatever --warmup says, they are over
        # before the timed steps
        for _ in range(max(0, engine.PLACE_DEEP_AT + 1 - (1 + args.warmup))):
            step_session(X)
    else:
        step_engine(X, H, out32, keep=True)
    dt, ev, host_t, own_dt = timed_steps(X, H, out32, args.steps, args.warmup, graph=args.graph,
                                         events=not os.environ.get("EPG_BENCH_NOEVENTS"), session=use_session)

    # sanity of the last step (cheap, outside the timed region): every state byte counted, scores finite
    table_note = None
    if use_session:
        sess = last["sess"]
        sess.finish(R_global, N)                # the job's host side: count check (sum == bins * biosamples), table verification
        patched = sess.tables_patched
        table_note = {"built_on": "device (k_s1_combine, in the launch that normalises)",
                      "float32_table_equals_numpy_reference_table": patched == 0}
        res = last["out"]
    else:
        total = int(last_counts.sum().item())
        assert total == R_global * N, "state counts %d != bins*biosamples %d" % (total, R_global * N)
        assert int(counts.abs().sum().item()) == 0                         # every job left the accumulator zeroed
        res = out32
    assert bool(torch.isfinite(res[:: max(R // 4096, 1)]).all())
    del res

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if not ev:                                                             # graph replay / no events: per-kernel times from a probe
        _, ev, _, _ = timed_steps(X, H, out32, min(args.steps, 10), 1, session=use_session)
    if os.environ.get("EPG_BENCH_TRACE") and rank == 0:      # per-step k_bin_hist times (clock ramps, box variance)
        print("k_bin_hist ms per step:", " ".join("%.3f" % e[0].elapsed_time(e[1]) for e in ev), file=sys.stderr, flush=True)
        print("host enqueue done at ms:", " ".join("%.1f" % (t * 1e3) for t in host_t), file=sys.stderr, flush=True)
    hist_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ar_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    rest_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))

    # ---- several ranks: what every rank saw (K1 per launch, its own time for the K steps), gathered on rank 0
    per_rank = None
    who = rank_identity(torch, dev, int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        whos = [None] * world
        try:
            dist.all_gather_object(whos, who)
        except Exception as e:                                             # identity is a courtesy: never lose the line for it
            whos = [who] + [{"error": repr(e)[:120]}] * (world - 1)
    else:
        whos = [who]
    if world > 1:
        mine = torch.tensor([hist_ms, own_dt * 1e3 / args.steps, ar_ms, rest_ms], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        allv = np.array([v.cpu().numpy() for v in allv])
        per_rank = {"k_bin_hist_ms": [round(float(v), 4) for v in allv[:, 0]],
                    "k_bin_hist_ms_min_max": [round(float(allv[:, 0].min()), 4), round(float(allv[:, 0].max()), 4)],
                    "own_ms_per_step": [round(float(v), 4) for v in allv[:, 1]],
                    "skew_ms_per_step": round(float(allv[:, 1].max() - allv[:, 1].min()), 4),
                    "allreduce_ms": [round(float(v), 4) for v in allv[:, 2]],
                    "combine+score_ms": [round(float(v), 4) for v in allv[:, 3]],
                    "ranks": whos,
                    "what": "own_ms_per_step = a rank's wall time from the common start until ITS last step had drained, over the "
                            "steps (the closing barrier excluded); skew = slowest - fastest rank"}
    elif use_pg or os.environ.get("EPG_BENCH_WHO"):
        per_rank = {"ranks": whos}

    # ---- one GPU: the other path on the same genome, and both paths on an eighth of it (the share of one of 8 GPUs)
    s1_paths = None
    if world == 1 and not args.graph:
        k = min(args.steps, 10)

        def both(Xs, k):
            # Each path twice, alternating, the better of the two (a 0.4 ms step is at the mercy of clock ramps and of whatever
            # the previous measurement left in the caches: single runs of ten steps scattered by +-5 %).  The bare-ABI path gets
            # its histogram cache and score buffer from the allocator right after a session run has returned its own, i.e. the
            # SAME blocks: where the driver happened to place H relative to X decides 17 % of K1, and two buffers
            # of one process can differ in that -- the comparison is about the code path, not about that draw.
            r = {}
            rows = Xs.shape[0]
            for name, sessn in (("session", True), ("engine", False), ("session_with_finish", True), ("session", True), ("engine", False),
                                ("session_with_finish", True)):
                last.clear()
                Hs = outs = None
                if not sessn:
                    Hs = engine.alloc_hist(Xs, N, S)                       # the home the session's jobs used (free again by now)
                    outs = torch.empty((rows, S), dtype=torch.float32, device=dev)
                # (warm-up of 4: engine.alloc_hist's searches, if this matrix still has them ahead, run at jobs 2 and 4 -- not in the timed steps)
                t, evs, _, _ = timed_steps(Xs, Hs, outs, k, 4, session=sessn, finish=name == "session_with_finish")
                del Hs, outs
                ms = round(t / k * 1e3, 4)
                if name + "_ms_per_step" not in r or ms < r[name + "_ms_per_step"]:
                    r[name + "_ms_per_step"] = ms
                    r[name + "_k_bin_hist_ms"] = round(float(np.mean([e[0].elapsed_time(e[1]) for e in evs])), 4)
            r["steps"] = k
            r["session_over_engine"] = round(r["session_ms_per_step"] / r["engine_ms_per_step"], 4)
            return r
        s1_paths = {"what": "session = backend._HipSingleSession per job (the product's calls and allocations: add_device, all_reduce, launch); "
                            "session_with_finish = the same plus finish() inside every job (count check, S1 table verification against "
                            "numpy, exp_freq download: the host synchronisation the command line has once per run); engine = bare ABI "
                            "calls, counts re-zeroed by the score launch, histogram cache from engine.alloc_hist as well (the same "
                            "home block: the comparison is about the code path, not about where H lies)",
                    "genome_%d_bins" % R: both(X, k)}
        rs = min(args.shard_bins, R)
        if 0 < rs < R:
            # the shard as a rank of an 8-GPU run holds it: a matrix of its own (its own allocation, its own placement decision), not a
            # view of the genome's
            Xshard = X[:rs].clone()
            s1_paths["shard_%d_bins" % rs] = both(Xshard, max(5 * k, 50))
            del Xshard
        last.clear()
        # the genome the way the COMMAND LINE holds it: 24 chromosome-sized parts (hg19's proportions), each a matrix of its own
        # (one upload per file), through the same session calls.  Parts under a GiB are counted and scored in BATCHES (8 M rows or
        # 32 parts per launch; histogram caches from one plain allocation per batch): two count launches, one combine, two score
        # launches per genome (round 6; a launch pair per part before: 3.42 ms)
        try:
            pieces = [X[a - bin0:b - bin0].clone() for _f, _r0, a, b in chromosome_parts(R_global, bin0, bin0 + R)]
            torch.cuda.synchronize()

            def job_parts():
                last.clear()
                sess = be.open_single(S, 1)
                pids = [sess.add_device(P, N) for P in pieces]
                sess.ensure_acc(N)
                sess.all_reduce(d)
                sess.launch(R_global, N, pids)
                last["sess"], last["out"] = sess, [sess.early_scores(p) for p in pids]
            for _ in range(3):
                job_parts()
            fence()
            t0 = time.perf_counter()
            for _ in range(k):
                job_parts()
            fence()
            tp = (time.perf_counter() - t0) / k
            last["sess"].finish(R_global, N)
            s1_paths["genome_as_%d_chromosome_parts" % len(pieces)] = {
                "ms_per_step": round(tp * 1e3, 4), "value": round(R / tp / 1e6, 3), "unit": "Mbins/s", "steps": k,
                "what": "the same S1 job with the genome held as the command line holds it: one resident matrix per chromosome file (%d), "
                        "counted and scored in batches of parts (one launch per 8 M rows / 32 parts), plain histogram caches" % len(pieces)}
            del pieces
            last.clear()
        except Exception as e:                                              # an extra: never let it take the measurement down
            s1_paths["genome_as_chromosome_parts"] = {"failed": repr(e)[:200]}

    # ---- the all-reduce by itself (device time between events, host time of the call), when there is a process group
    allreduce_probe = None
    if use_pg:
        t = torch.zeros(S, dtype=torch.int64, device=dev)
        for _ in range(20):
            dist.all_reduce(t)
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_ar = 200
        e0.record()
        th = time.perf_counter()
        for _ in range(n_ar):
            dist.all_reduce(t)
        host_us = (time.perf_counter() - th) / n_ar * 1e6
        e1.record()
        torch.cuda.synchronize()
        allreduce_probe = {"tensor": "int64[%d]" % S, "calls": n_ar, "device_us_per_call_back_to_back": round(e0.elapsed_time(e1) / n_ar * 1e3, 2),
                           "host_us_per_call": round(host_us, 2), "backend": args.backend, "world": world}

    # ---- the line: everything the timed region produced is in it from here on; what follows (the unplaced jobs, BASELINE
    # configs 3-5) fills `placement` and `configs` in.  A watchdog on every rank bounds those extras: past --extras-deadline
    # seconds rank 0 prints the line with what it has and every rank leaves, so that a collective that never completes in a
    # secondary measurement cannot take the headline of a multi-GPU run with it.
    placement = {"headline": ("a fresh backend._HipSingleSession per step; add_device's default hands the histogram cache of a resident matrix "
                              "of >= 1 GiB to engine.alloc_hist (another memory class than the matrix: a bounded walk over <= 8 blocks of 4 GiB, a "
                              "relative decision, found once per process during the warm-up; `report` is what it did, `search_ms` its wall time)"
                              if use_session else "bare ABI calls on preallocated plain torch allocations (--path engine / --graph)")}
    configs = {}
    line = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = R_global * args.steps / dt / 1e6
        achieved = R * N / (hist_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of tools/profile_bench.sh (rocprofv3 cannot run inside this process):
        # reported only when they were taken on this very kernel source and this shape
        traffic, traffic_note = None, "no PMC record for this shape"
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists():
            try:
                rec = json.loads(tfile.read_text())
                if rec.get("k1_source_sha") != k1_source_sha():
                    traffic_note = "PMC record is for other kernel source (%s), not reported" % rec.get("k1_source_sha")
                else:
                    traffic = rec.get("k_bin_hist_bytes_per_launch_%d_%d" % (R, N))
                    traffic_note = rec.get("source") if traffic is not None else traffic_note
            except Exception as e:
                traffic_note = "unreadable PMC record: %r" % (e,)
        line = {
            "metric": "Mbins scored/sec (S1, 18-state, 833 biosamples)",
            "value": round(value, 3), "unit": "Mbins/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "S1 saliency, whole-genome scale: %d bins x %d biosamples x %d states %s, "
                                   "expected pass + count all-reduce + normalise + score pass per step"
                                   % (args.bins, N, S, "split over the GPUs (splitRows)" if scaling == "strong" else "per GPU"),
                       "bins_total": R_global, "bins_per_gpu": R, "biosamples": N, "states": S, "saliency": 1,
                       "state_distribution": args.dist,
                       "row_pitch_bytes": int(X.stride(0)),
                       "step_launch": "hipGraph replay" if args.graph else "three launches + collective per step",
                       "step_path": ("backend._HipSingleSession: add_device(X, N) -> all_reduce -> launch -- the calls driver.run_single makes "
                                     "(its add_part is an upload followed by this same add_device(X, N), same default: engine.alloc_hist places the "
                                     "histogram cache of a resident matrix of >= 1 GiB, smaller ones get a plain allocation).  Here the genome is ONE "
                                     "resident 12.7 GB matrix, so its cache is placed; the command line holds it as one part per chromosome file "
                                     "(< 1 GiB each at 833 columns), i.e. plain allocations: `placement.unplaced` is the step with those and "
                                     "`s1_paths.genome_as_24_chromosome_parts` the job fed as those 24 parts.  The session's "
                                     "finish() (count check, table verification, exp_freq download: a host sync) runs once after the timed steps -- "
                                     "s1_paths.session_with_finish_ms_per_step has it inside every job") if use_session else "engine",
                       "partition": "contiguous bin ranges per GPU (helpers.splitRows rule), one RCCL all-reduce of int64[%d]" % S},
            "roofline": {"bound": "hbm", "kernel": "k_bin_hist", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_bin": N, "avg_launch_ms": round(hist_ms, 4),
                         # the whole step against the bytes it has to move: X read once, H written and read, float32 scores written
                         "step_bytes_per_bin": N + 2 * 2 * S + 4 * S,
                         "step_GBps": round(R * (N + 2 * 2 * S + 4 * S) / (ms_per_step * 1e-3) / 1e9, 1)},
            "kernels_ms": {"k_bin_hist": round(hist_ms, 4), "allreduce": round(ar_ms, 4),
                           "combine(normalise,table)+score_from_hist": round(rest_ms, 4)},
            "allreduce_probe": allreduce_probe,
            "per_rank": per_rank,
            "s1_paths": s1_paths,
            "s1_table": table_note,
            "placement": placement,
            "configs": configs,
            "cpu_baseline": cpu,
        }
    emitted = threading.Lock()

    def serialise(**more):
        for attempt in range(50):                          # the watchdog may serialise while the main thread adds a config
            try:
                return json.dumps(dict(line, **more))
            except RuntimeError:
                time.sleep(0.01)
        # never leave without the headline: drop what is still being filled in
        return json.dumps(dict(line, configs={"deadline": "extras dropped"}, placement=None, **more))

    def emit():
        if rank == 0 and emitted.acquire(blocking=False):
            text = serialise()
            if keeper is None or not keeper.final(text):
                sys.stdout.flush()
                os.write(real_stdout, (text + "\n").encode())    # (whatever a library has done to file descriptor 1 meanwhile)

    def checkpoint(leg):
        """More than one rank: the line as it stands goes to the keeper before every secondary measurement; it is what comes
        out if this process does not live to print the complete one."""
        if keeper is not None and rank == 0:
            keeper.keep(serialise(ended_early="rank 0 ended in the secondary measurement %r: the line is the state before it" % leg))

    def bail():
        # the headline is printed, then the process says what happened: exit code 3 (a hung collective in a secondary
        # measurement must not look like a clean run); torch.distributed.run then ends the other ranks
        configs["deadline"] = "extras stopped after %d s (--extras-deadline) in %r: exit code 3" % (args.extras_deadline, progress.get("leg"))
        emit()
        sys.stdout.flush()
        sys.stderr.flush()
        if rank != 0:
            time.sleep(3)                                  # (the launcher ends every rank when one fails: let rank 0 print first)
        os._exit(3)

    progress = {"leg": None}
    watchdog = None
    if args.extras_deadline > 0:
        watchdog = threading.Timer(args.extras_deadline, bail)
        watchdog.daemon = True
        watchdog.start()

    checkpoint("(none started)")

    def hang_hook(name):
        progress["leg"] = name
        checkpoint(name)
        if os.environ.get("EPG_BENCH_HANG_LEG") == name:               # tests: a secondary measurement that never returns
            time.sleep(10 ** 6)
        if os.environ.get("EPG_BENCH_ABORT_LEG") == name and rank == 0:  # tests: rank 0 dies in a secondary measurement
            os.abort()

    # ---- secondary: the same jobs with the histogram cache where a plain allocation puts it (add_device(place=False))
    placement["report"] = engine.placement_report(dev)
    if placement["report"]:
        placement["search_ms"] = placement["report"].get("search_ms")
        placement["left_in_torch_cache_GiB"] = placement["report"].get("left_in_torch_cache_GiB")
    if args.placement_experiment and world == 1 and R * X.stride(0) >= (1 << 30) and not args.packed and use_session:
        hang_hook("unplaced")
        try:
            kk = min(args.steps, 10)
            dtp, evp, _, _ = timed_steps(X, None, None, kk, 2, session=True, place=False)
            k1p = float(np.mean([e[0].elapsed_time(e[1]) for e in evp]))
            placement["unplaced"] = {"what": "the same jobs, histogram cache from a plain allocation (add_device(place=False): what the command "
                                             "line's < 1 GiB chromosome parts get under the default rule, and every job before round 5)", "steps": kk,
                                     "k_bin_hist_ms": round(k1p, 4), "frac": round(R * N / (k1p * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                     "ms_per_step": round(dtp / kk * 1e3, 4), "value": round(R_global * kk / dtp / 1e6, 3), "unit": "Mbins/s"}
            last.clear()
        except Exception as e:                 # an experiment: never let it take the measurement down
            placement["unplaced"] = {"failed": repr(e)[:200]}

    # ---- BASELINE configs 3-5 through the product's sessions
    want = [] if args.configs in ("none", "") else [c.strip() for c in args.configs.split(",") if c.strip()]
    if want:
        for name in want:
            hang_hook(name)
            try:
                if name == "s2":
                    configs["s2"] = bench_single(torch, be, d, 2, X, N, S, R, R_global, world, args.config_reps, fence)
                elif name in ("s3", "s3_small"):
                    r3g = min(args.s3_small_bins, R_global) if name == "s3_small" else min(args.s3_bins or R_global, R_global)
                    lo, hi = rank * r3g // world, (rank + 1) * r3g // world
                    reps3 = args.config_reps if name == "s3_small" else 1
                    configs[name] = bench_single(torch, be, d, 3, X[: hi - lo], N, S, hi - lo, r3g, world, reps3, fence)
                elif name == "paired":
                    configs["paired"] = bench_paired(torch, be, d, R, R_global, bin0, S, world, args.config_reps, fence, dev, args.dist)
                else:
                    configs[name] = {"error": "unknown config"}
            except Exception as e:
                configs[name] = {"error": repr(e)[:300]}

    # ---- SURVEY 8d's distribution variants: the same job on a uniform matrix (the contention-free control) and on the
    # row-correlated stress shape, generated into the SAME buffer (the headline's matrix is not needed any more)
    if args.dist_variants and world == 1 and use_session and args.dist == "chr1" and line is not None:
        hang_hook("dist_variants")
        try:
            dv = {"chr1 (headline)": {"k_bin_hist_ms": round(hist_ms, 4), "frac": round(R * N / (hist_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                      "ms_per_step": round(dt / args.steps * 1e3, 4)}}
            for name in ("uniform", "correlated", "chr1"):
                generate_shard(torch, X, N, S, bin0, dist=name)
                dtv, evv, _, _ = timed_steps(X, None, None, 5, 2, session=True)
                last["sess"].finish(R_global, N)
                k1v = float(np.mean([e[0].elapsed_time(e[1]) for e in evv]))
                dv[name if name != "chr1" else "chr1 (again, after the variants)"] = {
                    "k_bin_hist_ms": round(k1v, 4), "frac": round(R * N / (k1v * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "ms_per_step": round(dtv / 5 * 1e3, 4),
                    "modal_state_share": round(float((X[:: max(R // 4096, 1), :N] == S - 1).float().mean().item()), 4)}
                last.clear()
            dv["what"] = ("5 whole S1 jobs each through the session on the same buffer (same placement): uniform states, and the "
                          "row-correlated shape (41 % of the bins all-quiescent, 83 % of the rest copies of their predecessor); the counting "
                          "core has no data-dependent control flow and no shared counters (csrc/epg_count.h)")
            line["dist_variants"] = dv
        except Exception as e:
            line["dist_variants"] = {"error": repr(e)[:300]}

    # ---- RCCL self-test: every collective of the multi-rank command line against numbers (reported, never fatal)
    if use_pg:
        hang_hook("rccl_selftest")
        st = rccl_selftest(torch, dist, dev, rank, world, N, S)
        oks = torch.tensor([1 if st["ok"] else 0], dtype=torch.int32, device=dev)
        try:
            dist.all_reduce(oks, op=dist.ReduceOp.MIN)
            st["ok_on_every_rank"] = bool(int(oks.item()))
        except Exception as e:
            st["ok_on_every_rank"] = False
            st["error"] = repr(e)[:200]
        if line is not None:
            line["rccl_selftest"] = st

    # ---- LAST (a captured collective is the one leg no box of this pool could try with more than one rank): the same step as ONE
    # hipGraph replay (K1, all-reduce, combine, score captured together): what is left of the step when
    # the host side of three launches and a collective is taken out -- on an eighth of the genome that is 10 % of the step
    if args.graph_leg and not args.graph and use_pg and args.backend != "nccl":
        if line is not None:                                             # (a host-side backend syncs inside the collective)
            line["graph_ms_per_step"], line["graph_error"] = None, "not captured: the %s backend's all-reduce cannot be part of a hipGraph" % args.backend
    elif args.graph_leg and not args.graph:
        hang_hook("graph")
        try:
            Hg = engine.alloc_hist(X, N, S)
            og = torch.empty((R, S), dtype=torch.float32, device=dev)
            dtg, _, _, _ = timed_steps(X, Hg, og, args.steps, args.warmup, graph=True)
            dtg = _max_over_ranks(torch, d, dtg, dev)
            if line is not None:
                line["graph_ms_per_step"] = round(dtg / args.steps * 1e3, 4)
                line["graph_value"] = round(R_global * args.steps / dtg / 1e6, 3)
            del Hg, og
        except Exception as e:
            if line is not None:
                line["graph_ms_per_step"] = None
                line["graph_error"] = repr(e)[:300]

    # ---- a prediction for the first real 8-GPU run to be held against (one GPU only): the genome step over the step of one
    # GPU's share of it plus the device time of an RCCL all-reduce of the count vector -- measured here in a ONE-rank group (the
    # collective's launch and kernel; the xGMI hops of seven more ranks are NOT in it) --; rank skew is not in it either
    if world == 1 and line is not None and s1_paths and any(k.startswith("shard_") for k in s1_paths):
        hang_hook("projected_speedup")
        try:
            ar_us, ar_src = None, None
            if allreduce_probe is not None:
                ar_us, ar_src = allreduce_probe["device_us_per_call_back_to_back"], "this run's one-rank group (--pg)"
            elif args.allreduce_leg:
                import subprocess
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
                try:
                    res = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--allreduce-child", str(S)], env=env, capture_output=True,
                                         text=True, timeout=90)
                    got = [l for l in res.stdout.splitlines() if l.startswith("{")]
                    if res.returncode == 0 and got:
                        ar_us, ar_src = json.loads(got[-1])["allreduce_us"], "a one-rank RCCL group in a child process, after the timed region"
                    else:
                        ar_src = "the child process failed (rc %s): %s" % (res.returncode, (res.stderr or "")[-160:])
                except subprocess.TimeoutExpired:
                    ar_src = "the child process did not bring a one-rank RCCL group up within 90 s"
            gk = [k for k in s1_paths if k.startswith("genome_")][0]
            sk = [k for k in s1_paths if k.startswith("shard_")][0]
            g_ms, s_ms = s1_paths[gk]["session_ms_per_step"], s1_paths[sk]["session_ms_per_step"]
            line["projected_speedup_8"] = {
                "value": round(g_ms / (s_ms + (ar_us if ar_us is not None else 9.6) / 1e3), 2), "genome_step_ms": g_ms, "shard_step_ms": s_ms,
                "allreduce_us_used": ar_us if ar_us is not None else 9.6,     # (9.5-9.7 us on every box that measured it)
                "shard_bins": int(sk.split("_")[1]), "allreduce_us": ar_us, "allreduce_source": ar_src,
                "shard_k_bin_hist_frac": round(int(sk.split("_")[1]) * N / (s1_paths[sk]["session_k_bin_hist_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "shard_after_k_bin_hist_us": round((s_ms - s1_paths[sk]["session_k_bin_hist_ms"]) * 1e3, 1),
                "what": "genome step / (step of one of 8 shards + one all-reduce of int64[%d]), all on this one GPU: what the 8-GPU run can "
                        "reach before rank skew and the hops of a real 8-rank ring; north_star asks for >= 6" % S}
        except Exception as e:
            line["projected_speedup_8"] = {"error": repr(e)[:300]}

    if watchdog is not None:
        watchdog.cancel()
    emit()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

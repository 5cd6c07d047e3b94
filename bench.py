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
"""
import argparse
import json
import os
import sys
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


def generate_shard(torch, X, n_biosamples, n_states, bin0, dist="chr1"):
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
        gen.manual_seed(1234 + k)
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


def main():
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
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to test the "
                                                      "multi-rank path on a box with one GPU, ranks then share cuda:0)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    # ---- resident inputs and preallocated outputs
    if args.packed:
        flat = torch.empty(R * N + 64, dtype=torch.int8, device=dev)
        X = flat[:R * N].view(R, N)
    else:
        X = engine.alloc_states(R, N, device=dev)
    generate_shard(torch, X, N, S, bin0, dist=args.dist)
    counts = torch.zeros(S, dtype=torch.int64, device=dev)
    q = torch.empty(S, dtype=torch.float32, device=dev)
    out32 = torch.empty((R, S), dtype=torch.float32, device=dev)
    ws_s1 = engine.workspace(1, 0, N, S, device=dev)
    # last, the histogram cache: it goes where k_bin_hist's writes do not collide with its reads of X (engine.place_hist: the
    # device memory has three classes of regions, X and H in one class = 17 % slower); what was tried is in the JSON line
    try:
        H, placement = engine.place_hist(X, N, S, park=True)   # the blocks it did not keep go back after the timed region
    except Exception as e:                     # the search is an optimisation: never let it take the measurement down
        torch.cuda.empty_cache()
        H, placement = torch.empty((R, S), dtype=torch.int16, device=dev), {"tries": 1, "search_failed": repr(e)[:200]}
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    last_counts = torch.zeros(S, dtype=torch.int64, device=dev)

    if os.environ.get("EPG_BENCH_NOEVENTS"):         # diagnostic: what the three event records per step cost (kernels_ms is then NaN)
        ev = []

    def step(k=None, keep=False):
        if not ev:
            k = None
        if k is not None:
            ev[k][0].record()
        engine.bin_hist(X, N, S, counts=counts, H=H)                       # STEP 1: expected pass (counts start zeroed)
        if k is not None:
            ev[k][1].record()
        if world > 1:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)                  # the single collective (144 bytes)
        if keep:
            last_counts.copy_(counts)                                      # outside the timed steps: for the sanity check
        # STEP 2 + STEP 3: normalise + table in one single-block kernel (it leaves counts zeroed for the next job), score pass
        engine.combine_score_s1(counts, H, N, S, q=q, out32=out32, ws=ws_s1, rezero=True)
        if k is not None:
            ev[k][2].record()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step(keep=True)                                                        # untimed: sanity of one whole job
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    host_t = []
    for k in range(args.steps):
        step(k)
        host_t.append(time.perf_counter() - t0)
    fence()
    dt = time.perf_counter() - t0

    engine.release_parked()
    # sanity of the last step (cheap, outside the timed region): every state byte counted, scores finite
    total = int(last_counts.sum().item())
    assert total == R_global * N, "state counts %d != bins*biosamples %d" % (total, R_global * N)
    assert int(counts.abs().sum().item()) == 0                             # every job left the accumulator zeroed
    assert bool(torch.isfinite(out32[:: max(R // 4096, 1)]).all())

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if os.environ.get("EPG_BENCH_TRACE") and rank == 0:      # per-step k_bin_hist times (clock ramps, box variance)
        print("k_bin_hist ms per step:", " ".join("%.3f" % e[0].elapsed_time(e[1]) for e in ev), file=sys.stderr, flush=True)
        print("host enqueue done at ms:", " ".join("%.1f" % (t * 1e3) for t in host_t), file=sys.stderr, flush=True)
    hist_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) if ev else float("nan")
    rest_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if ev else float("nan")

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
                       "partition": "contiguous bin ranges per GPU (helpers.splitRows rule), one RCCL all-reduce of int64[%d]" % S},
            "roofline": {"bound": "hbm", "kernel": "k_bin_hist", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_bin": N, "avg_launch_ms": round(hist_ms, 4),
                         # the whole step against the bytes it has to move: X read once, H written and read, float32 scores written
                         "step_bytes_per_bin": N + 2 * 2 * S + 4 * S,
                         "step_GBps": round(R * (N + 2 * 2 * S + 4 * S) / (ms_per_step * 1e-3) / 1e9, 1)},
            "kernels_ms": {"k_bin_hist": round(hist_ms, 4), "allreduce+combine(normalise,table)+score_from_hist": round(rest_ms, 4)},
            "placement": placement,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

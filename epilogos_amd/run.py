"""`epilogos` command line for the MI355X engine: the option surface of the reference's epilogos/run.py
(click options :18-73, checkFlags :328-375, checkArguments :378-451, fileTag/paths :158-165) for STEP 1-3.
The per-chromosome SLURM submission (run.py:454-585) is replaced by a bin-range partition across the GPUs of the
node: `--gpus N` starts one process per GPU (a child `python -m torch.distributed.run`, started before this process
touches a GPU) the way the reference's one command fans out over SLURM by itself; a process that finds itself under
torch.distributed.run (WORLD_SIZE set) joins the group.  Console script: `epilogos` (pyproject.toml -> run:cli).  Single mode
also runs STEP 4 (regionsOfInterest_*.txt, epilogos_amd/roiSingle.py); paired mode runs the STEP 4 of
epilogos_amd/roiAndVisualPairwise.py (pairwiseMetrics, regionsOfInterest, significantLoci; no figures)."""
import os
import re
import sys
import time as _time
from pathlib import Path, PurePath

_T_START = _time.time()

import click

from . import __version__
from .helpers import getNumStates


def _natural_key(p):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", Path(p).name)]


def checkArguments(mode, saliency, inputDirPath, inputDirPath2, outputDirPath, numProcesses, numStates, quiescentState,
                   groupSize, numTrials=101, samplingSize=100000, roiWidth=0):
    """Reference run.py:378-451; same exceptions and messages."""
    if mode == "paired" and saliency == 3:
        raise ValueError("Paired epilogos supports only a saliency of 1 or 2")
    if saliency not in (1, 2, 3):
        raise ValueError("Please ensure that saliency metric is either 1, 2, or 3")
    for d in [inputDirPath] + ([inputDirPath2] if mode == "paired" else []):
        if not d.exists():
            raise FileNotFoundError("Given path does not exist: {}".format(str(d)))
        if not d.is_dir():
            raise NotADirectoryError("Given path is not a directory: {}".format(str(d)))
        if not list(d.glob("*")):
            raise OSError("Ensure that directory is not empty: {}".format(str(d)))
    if outputDirPath == inputDirPath:
        print("ERROR: Output directory cannot be the same as the input directory")
        sys.exit()
    if not outputDirPath.exists():
        outputDirPath.mkdir(parents=True, exist_ok=True)       # every rank gets here; the first one creates it
    if not outputDirPath.is_dir():
        raise NotADirectoryError("Given path is not a directory: {}".format(str(outputDirPath)))
    if numProcesses < 0:
        print("ERROR: Number of cores must be positive or zero (0 means use all cores)")
        sys.exit()
    if numTrials <= 0:
        print("ERROR: Number of trials must be greater than zero")
        sys.exit()
    if samplingSize <= 0:
        print("ERROR: Sampling size must be greater than zero")
        sys.exit()
    if roiWidth < 0:
        print("ERROR: Group size value must be greater than 0")
        sys.exit()
    if quiescentState >= numStates:
        print("ERROR: Quiescent state must be a valid state (at most the number of states)")
        sys.exit()
    if groupSize < -1 or groupSize == 0:
        print("ERROR: Group size must be positive")
        sys.exit()


def _init_device_and_group(world, under_launcher):
    """Import torch, pick this rank's GPU and join the process group (also a group of one: the collective path is the same code).
    -> torch.device or None."""
    import torch
    device = None
    # EPILOGOS_DIST_BACKEND=gloo lets several ranks share one GPU (testing the multi-rank path on a 1-GPU box)
    dist_backend = os.environ.get("EPILOGOS_DIST_BACKEND", "")
    if torch.cuda.is_available():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if not dist_backend and local >= torch.cuda.device_count():
            raise SystemExit("ERROR: rank %d of this node has no GPU: %d rank(s) were started for %d usable device(s) "
                             "(--gpus / *_VISIBLE_DEVICES)" % (local, int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0), torch.cuda.device_count()))
        torch.cuda.set_device(local % torch.cuda.device_count() if dist_backend else local)
        device = torch.device("cuda", torch.cuda.current_device())
    if world > 1 or under_launcher:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = dist_backend or ("nccl" if device is not None else "gloo")
        # a collective whose partner never arrives (a rank that died where the launcher cannot see it) must end the job, not
        # hold seven GPUs forever; generous, because a rank legitimately waits while the others still parse their files
        import datetime
        limit = datetime.timedelta(seconds=int(os.environ.get("EPILOGOS_DIST_TIMEOUT", "1800")))
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device, timeout=limit)
        else:
            dist.init_process_group(backend=backend, timeout=limit)
    return device


@click.command(context_settings=dict(help_option_names=["-h", "--help"]))
@click.option("-m", "--mode", "mode", type=click.Choice(["single", "paired"]), default="single", show_default=True,
              help="single for single group epilogos and paired for 2 group epilogos")
@click.option("-l", "--local", "commandLineBool", is_flag=True,
              help="Run in this process (always the case for the GPU engine; accepted for compatibility)")
@click.option("-i", "--input-directory", "inputDirectory", type=str, help="Directory with one matrix file per chromosome")
@click.option("-a", "--directory-one", "inputDirectory1", type=str, help="First input directory (paired mode)")
@click.option("-b", "--directory-two", "inputDirectory2", type=str, help="Second input directory (paired mode)")
@click.option("-o", "--output-directory", "outputDirectory", type=str, help="Output directory")
@click.option("-j", "--state-info", "stateInfo", type=str, help="State model info file")
@click.option("-s", "--saliency", "saliency", type=int, default=1, show_default=True, help="Saliency level (1, 2, or 3)")
@click.option("-c", "--num-cores", "numProcesses", type=int, default=0, show_default=True,
              help="Upper bound on the host cores the job uses for parsing, writing and STEP 4 (0 = all it may use); shared out "
                   "over the ranks of --gpus")
@click.option("-x", "--exit", "exitBool", is_flag=True, help="SLURM-only flag; accepted and ignored")
@click.option("-d", "--diagnostic-figures", "diagnosticBool", is_flag=True, help="Figures are not produced; accepted and ignored")
@click.option("-t", "--num-trials", "numTrials", type=int, default=101, show_default=True,
              help="Number of gennorm fits of the null distances (paired mode with -n)")
@click.option("-z", "--sampling-size", "samplingSize", type=int, default=100000, show_default=True,
              help="Size of the null sub-sample of each fit (paired mode with -n)")
@click.option("-q", "--quiescent-state", "quiescentState", type=int, default=-1,
              help="1-based quiescent state for paired filtering; 0 disables it [default: last state]")
@click.option("-g", "--group-size", "groupSize", type=int, default=-1, show_default=True,
              help="Size of the shuffled null groups in paired mode (default: the input group sizes)")
@click.option("-v", "--version", "version", is_flag=True, help="Print the version and exit")
@click.option("-p", "--partition", "partition", type=str, help="SLURM-only flag; accepted and ignored")
@click.option("-n", "--null-distribution", "pvalBool", is_flag=True,
              help="Paired mode: fit the null distances and report p-values instead of z-scores")
@click.option("-w", "--roi-width", "roiWidth", type=int, default=0, help="Bins per region of interest [default: 50 single, 125 paired]")
@click.option("-f", "--file-tag", "fileTag", type=str, default="null",
              help="Tag appended to output filenames [default: input-directory_saliency]")
@click.option("--exp-freq-mem", "expFreqMem", type=int, default=20000, help="SLURM-only; ignored")
@click.option("--exp-comb-mem", "expCombMem", type=int, default=8000, help="SLURM-only; ignored")
@click.option("--score-mem", "scoreMem", type=int, default=40000, help="SLURM-only; ignored")
@click.option("--roi-mem", "roiMem", type=int, default=-1, help="SLURM-only; ignored")
@click.option("--null-seed", "nullSeed", type=int, default=None, help="Seed for the paired-mode null shuffles (default: random)")
@click.option("--gpus", "gpus", type=int, default=1, show_default=True,
              help="GPUs of this node to split the genome over (0 = all visible); replaces the SLURM fan-out of the reference")
@click.option("--cache-dir", "cacheDir", type=str, default=None,
              help="Keep parsed input matrices (int8 + coordinates) here; later runs on the same files skip the text parse")
def main(mode, commandLineBool, inputDirectory, inputDirectory1, inputDirectory2, outputDirectory, stateInfo, saliency,
         numProcesses, exitBool, diagnosticBool, numTrials, samplingSize, quiescentState, groupSize, version, partition,
         pvalBool, roiWidth, fileTag, expFreqMem, expCombMem, scoreMem, roiMem, nullSeed, gpus, cacheDir):
    """Information-theoretic navigation of multi-tissue functional genomic annotations -- MI355X scoring engine."""
    if version:
        print("Version:", __version__)
        sys.exit()
    # flag combinations (reference run.py:328-375)
    if mode == "single" and inputDirectory is None:
        print("ERROR: [-i, --input-directory] is required in single mode"); sys.exit()
    if mode == "single" and (inputDirectory1 is not None or inputDirectory2 is not None):
        print("ERROR: [-a] and [-b] are only valid in paired mode"); sys.exit()
    if mode == "paired" and (inputDirectory1 is None or inputDirectory2 is None):
        print("ERROR: [-a, --directory-one] and [-b, --directory-two] are required in paired mode"); sys.exit()
    if mode == "paired" and inputDirectory is not None:
        print("ERROR: [-i] is only valid in single mode"); sys.exit()
    if outputDirectory is None:
        print("ERROR: [-o, --output-directory] is required"); sys.exit()
    if stateInfo is None:
        print("ERROR: [-j, --state-info] is required"); sys.exit()

    if cacheDir:
        os.environ["EPILOGOS_CACHE_DIR"] = str(Path(cacheDir).resolve())
    if numProcesses > 0:                                      # the reference's core budget (run.py:36,148): see _io.host_budget
        os.environ["EPILOGOS_NUM_CORES"] = str(numProcesses)
    numStates = getNumStates(stateInfo)
    quiescentState = numStates - 1 if quiescentState == -1 else quiescentState - 1     # 1-based -> 0-based, 0 -> off
    inputDirPath = Path(inputDirectory if mode == "single" else inputDirectory1)
    inputDirPath2 = Path(inputDirectory2) if mode == "paired" else Path("")
    if not PurePath(inputDirPath).is_absolute():
        inputDirPath = Path.cwd() / inputDirPath
    if mode == "paired" and not PurePath(inputDirPath2).is_absolute():
        inputDirPath2 = Path.cwd() / inputDirPath2
    outputDirPath = Path(outputDirectory)
    if not PurePath(outputDirPath).is_absolute():
        outputDirPath = Path.cwd() / outputDirPath
    checkArguments(mode, saliency, inputDirPath, inputDirPath2, outputDirPath, numProcesses, numStates, quiescentState,
                   groupSize, numTrials, samplingSize, roiWidth)
    if fileTag == "null":
        fileTag = ("{}_s{}".format(inputDirPath.name, saliency) if mode == "single"
                   else "{}_{}_s{}".format(inputDirPath.name, inputDirPath2.name, saliency))
    storedExpPath = outputDirPath / "exp_freq_{}.npy".format(fileTag)

    # --gpus N from a plain start: fan out into one process per GPU (the reference submits its own SLURM jobs,
    # run.py:190-279,454-505) -- as a CHILD process, before this one has imported torch or touched a GPU
    if gpus < 0:
        print("ERROR: Number of GPUs must be positive or zero (0 means use all visible GPUs)"); sys.exit()
    if "WORLD_SIZE" not in os.environ and gpus != 1:
        if gpus == 0:
            gpus = _visible_gpus()
        if gpus > 1:
            sys.exit(_launch(gpus, sys.argv[1:] if _ARGV is None else _ARGV))

    # one process per GPU when launched under torch.distributed.run
    world = int(os.environ.get("WORLD_SIZE", "1"))
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    files = sorted(inputDirPath.glob("*"), key=_natural_key)
    files2 = []
    if mode == "paired":
        for file in files:
            if not list(inputDirPath2.glob(file.name)):
                raise FileNotFoundError("File not found: {}".format(str(inputDirPath2 / file.name))
                                        + " Please ensure corresponding files within input directories 1 and 2 have the same name")
            files2.append(next(inputDirPath2.glob(file.name)))
    from . import driver, _io
    if world == 1 and not under_launcher and os.environ.get("EPILOGOS_EARLY_READERS", "1") != "0":
        # A single process: the files' readers start NOW -- inflating needs neither torch nor the GPU, and importing the one and
        # initialising the other takes about a second, which a whole-genome run spent before its first byte was read.  A reader
        # asks for its (page-locked) destination only after its inflate; the session is attached when it exists.
        _io.set_state_limit(numStates)
        jobs = []
        for k, f in enumerate(files):
            jobs += [(f, 0, None)] + ([(files2[k], 0, None)] if mode == "paired" else [])
        driver.start_readers_early(jobs)
    try:
        device = _init_device_and_group(world, under_launcher)
    except BaseException:
        driver.abort_early_readers()
        raise
    rank = int(os.environ.get("RANK", "0"))
    say = print if rank == 0 else (lambda *a, **k: None)
    if os.environ.get("EPILOGOS_TIMING") and rank == 0:
        import time
        print("    [timing] %-34s %7.2f s" % ("imports + device init (since process start)", time.time() - _T_START), flush=True)

    say("State Model =", numStates, " Saliency level =", saliency, " GPUs =", world)
    if mode == "single":
        from .driver import run_single_group
        say("\nSTEP 1-3: background counts -> all-reduce -> scores (bin-range partition over %d GPU(s))" % world)
        try:
            _, results = run_single_group(files, numStates, saliency, outputDirPath, fileTag, verbose=False, device=device,
                                          keep_temp_scores=False, defer_writes=True)
        finally:
            driver.abort_early_readers()                         # (nothing to do once the stage driver has taken them over)
        try:
            if rank == 0:
                say("\nSTEP 4: Finding regions of interest", flush=True)
                import time
                t0 = time.perf_counter()
                from .roiSingle import mainFromArrays as roiSingle
                roiSingle(results, outputDirPath, stateInfo, fileTag, storedExpPath, roiWidth if roiWidth else 50, False)
                if os.environ.get("EPILOGOS_TIMING"):
                    print("    [timing] %-34s %7.2f s" % ("STEP 4", time.perf_counter() - t0), flush=True)
        finally:
            driver.finish_writes()                               # one process: the score text was still being written under STEP 4
    else:
        from .driver import run_paired_groups
        if nullSeed is None:
            import numpy as np
            seed = np.array([int(np.random.SeedSequence().generate_state(1)[0])], dtype=np.int64)
            if world > 1:                                        # every rank must shuffle with the same seed
                import torch
                import torch.distributed as dist
                t = torch.from_numpy(seed)
                t = t.to(device) if device is not None else t
                dist.broadcast(t, src=0)
                seed = t.cpu().numpy()
            nullSeed = int(seed[0])
        say("\nSTEP 1-3: background counts over [A|B] -> all-reduce -> scores, null groups, deltas (%d GPU(s))" % world)
        try:
            _, results = run_paired_groups(files, files2, numStates, saliency, outputDirPath, fileTag, quiescentState, groupSize,
                                           nullSeed, verbose=False, device=device, keep_temps=False, defer_writes=True)
        finally:
            driver.abort_early_readers()
        if pvalBool and max(numProcesses, 1) > 1:
            driver.finish_writes()                               # -n -c K forks a pool for the fits: not with writer threads running
        try:
            if rank == 0:
                say("\nSTEP 4: Generating p-values & regions of interest (figures are not produced)", flush=True)
                import time
                t0 = time.perf_counter()
                from .roiAndVisualPairwise import mainFromArrays as roiPairwise
                roiPairwise(results, stateInfo, outputDirPath, fileTag, max(numProcesses, 1), pvalBool, numTrials, samplingSize,
                            storedExpPath, roiWidth if roiWidth else 125, False)
                if os.environ.get("EPILOGOS_TIMING"):
                    print("    [timing] %-34s %7.2f s" % ("STEP 4", time.perf_counter() - t0), flush=True)
        finally:
            driver.finish_writes()
    from .helpers import flushCacheWrites
    flushCacheWrites()                                           # --cache-dir: files of first-time reads, written in the background
    if os.environ.get("EPILOGOS_TIMING") and rank == 0:
        import time
        print("    [timing] %-34s %7.2f s" % ("end of main (since process start)", time.time() - _T_START), flush=True)
        try:                                                     # what the process holds (and will have to give back at exit)
            st = dict(l.split(":", 1) for l in open("/proc/self/status").read().splitlines() if ":" in l)
            print("    [timing] memory at the end: " + ", ".join("%s %.1f GB" % (k, int(st[k].split()[0]) / 1048576.0)
                                                              for k in ("VmHWM", "VmRSS", "RssAnon", "RssFile", "RssShmem") if k in st), flush=True)
        except (OSError, ValueError):
            pass
    if world > 1 or under_launcher:
        import torch.distributed as dist
        dist.destroy_process_group()


_ARGV = None            # the argument list cli() was called with, when it is not sys.argv[1:]


def _visible_gpus():
    """GPUs this process could use, counted WITHOUT loading torch or touching HIP (the parent of the ranks must stay
    GPU-free): the KFD topology lists every compute node (a GPU has simd_count > 0, a CPU node 0), and the usual
    *_VISIBLE_DEVICES lists narrow it down."""
    n = 0
    try:
        for node in sorted(Path("/sys/class/kfd/kfd/topology/nodes").iterdir()):
            props = dict(l.split()[:2] for l in (node / "properties").read_text().splitlines() if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        n = 0
    # a container may see the host's whole KFD topology and only some of its render nodes: a GPU without an accessible
    # /dev/dri/renderD* cannot be opened (a rank on it would die in torch.cuda.set_device)
    try:
        usable = len([d for d in Path("/dev/dri").iterdir() if d.name.startswith("renderD") and os.access(d, os.R_OK | os.W_OK)])
        if n and usable:
            n = min(n, usable)
    except OSError:
        pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([t for t in v.split(",") if t.strip()])
            n = min(n, listed) if n else listed
    return max(n, 1)


def _strip_gpus(argv):
    """The argument list without --gpus (the children learn their number from WORLD_SIZE)."""
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
        elif a == "--gpus":
            skip = True
        elif not a.startswith("--gpus="):
            out.append(a)
    return out


def _launch_command(gpus, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "epilogos_amd.run"] + _strip_gpus(argv)


def _launch(gpus, argv):
    """One process per GPU under torch.distributed.run, as a child of this (GPU-free) process; its exit code is ours."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    root = str(Path(__file__).resolve().parents[1])
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    cmd = _launch_command(gpus, argv, port)
    if os.environ.get("EPILOGOS_LAUNCH_DRYRUN"):
        print(" ".join(cmd))
        return 0
    return subprocess.call(cmd, env=env)


def cli(argv=None):
    """Entry point of the `epilogos` console script and of `python -m epilogos_amd.run`."""
    global _ARGV
    _ARGV = argv
    try:
        main(args=argv, standalone_mode=False)
    except click.exceptions.Abort:
        print("Aborted!", file=sys.stderr)
        _code = 1
    except click.ClickException as e:
        e.show()
        _code = e.exit_code
    except SystemExit as e:                                  # the reference's `print("ERROR ..."); sys.exit()` paths
        _code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    else:
        _code = 0
    # Everything is written, flushed and closed at this point.  Tearing the interpreter down -- the HIP context, 13 GB of
    # pinned and mapped host memory, torch's module state -- took 0.7 s of a 4.5 s whole-genome run: leave at once.
    # (EPILOGOS_FAST_EXIT=0 restores the ordinary exit, e.g. under a coverage tool.)
    try:                                                     # error exits too: let the --cache-dir writers finish their files
        from .helpers import flushCacheWrites
        flushCacheWrites()
    except Exception:
        pass
    sys.stdout.flush()
    sys.stderr.flush()
    if os.environ.get("EPILOGOS_FAST_EXIT", "1") != "0":
        os._exit(_code)
    sys.exit(_code)


if __name__ == "__main__":
    cli()

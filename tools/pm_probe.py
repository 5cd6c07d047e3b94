#!/usr/bin/env python3
"""Does power management explain the two K1 levels inside one bench run?  A thread samples the GPU's sysfs clocks / power /
temperature every millisecond while 80 S1 steps run; the per-step K1 times are printed next to the samples."""
import glob
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from epilogos_amd import engine  # noqa: E402

devs = sorted(glob.glob("/sys/class/drm/card*/device"))
files = []
for d in devs:
    for pat in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "hwmon/hwmon*/power1_average", "hwmon/hwmon*/power1_input",
                "hwmon/hwmon*/freq1_input", "hwmon/hwmon*/freq2_input", "hwmon/hwmon*/temp1_input", "hwmon/hwmon*/temp2_input",
                "hwmon/hwmon*/temp3_input", "gpu_busy_percent", "mem_busy_percent"):
        files += glob.glob(d + "/" + pat)
ALL_FILES = files


def read(f):
    try:
        s = Path(f).read_text()
    except OSError as e:
        return "ERR %s" % e.errno
    if "pp_dpm" in f:
        cur = [ln for ln in s.splitlines() if ln.endswith("*")]
        return cur[0] if cur else s.replace("\n", "|")
    return s.strip()


N, S, R = 833, 18, 15_000_000
X = engine.alloc_states(R, N)
bench.generate_shard(torch, X, N, S, 0)
counts = torch.zeros(S, dtype=torch.int64, device="cuda")
q = torch.empty(S, dtype=torch.float32, device="cuda")
out32 = torch.empty((R, S), dtype=torch.float32, device="cuda")
ws = engine.workspace(1, 0, N, S)
H, rep = engine.place_hist(X, N, S)
print("placement", rep, flush=True)
STEPS = 80
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(STEPS)]
samples, stop = [], False


def sampler():
    while not stop:
        t = time.perf_counter()
        samples.append((t, [read(f) for f in files]))
        time.sleep(0.0005)


# which card is ours: power before / during a burn of K1 launches
torch.cuda.synchronize()
time.sleep(0.3)
pw = [f for f in ALL_FILES if f.endswith("power1_input")]
before = [int(Path(f).read_text()) for f in pw]
for _ in range(60):
    engine.bin_hist(X, N, S, counts=counts, H=H)
time.sleep(0.1)
during = [int(Path(f).read_text()) for f in pw]
torch.cuda.synchronize()
gain = [d - b for b, d in zip(before, during)]
mine = pw[gain.index(max(gain))].split("/device/")[0]
print("power before/during burn (W):", [(b // 1000000, d // 1000000) for b, d in zip(before, during)], "-> ours is", mine, flush=True)
files = [f for f in ALL_FILES if f.startswith(mine + "/")]
print("sysfs files:", [f.replace("/sys/class/drm/", "") for f in files], flush=True)
counts.zero_()
time.sleep(0.5)
th = threading.Thread(target=sampler)
th.start()
time.sleep(0.05)
t0 = time.perf_counter()
for k in range(STEPS):
    ev[k][0].record()
    engine.bin_hist(X, N, S, counts=counts, H=H)
    ev[k][1].record()
    engine.combine_score_s1(counts, H, N, S, q=q, out32=out32, ws=ws, rezero=True)
torch.cuda.synchronize()
t1 = time.perf_counter()
time.sleep(0.05)
stop = True
th.join()
k1 = [e[0].elapsed_time(e[1]) for e in ev]
print("K1 per step:", " ".join("%.3f" % v for v in k1))
print("run %.1f ms; samples (ms since start of the run: values), changes only:" % ((t1 - t0) * 1e3))
last = None
for t, v in samples:
    key = [x for x, f in zip(v, files) if "power" not in f and "temp" not in f and "busy" not in f]
    if key != last:
        print("  %7.1f: %s" % ((t - t0) * 1e3, v))
        last = key
print("  power/temp trace (every 8th sample):")
for t, v in samples[::4]:
    print("  %7.1f: %s" % ((t - t0) * 1e3, [x for x, f in zip(v, files) if "power" in f or "temp" in f or "busy" in f]))

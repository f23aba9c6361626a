#!/usr/bin/env python3
"""Shader clock, memory clock and package power (sysfs, 10 ms apart) while a command runs - does a kernel's
time move with the clocks?
    python tools/probe_clocks.py -- python bench.py --workload cfg3 --steps 400 ..."""
import glob
import statistics
import subprocess
import sys
import time

cmd = sys.argv[sys.argv.index("--") + 1:]


def find(pattern):
    hits = sorted(glob.glob(pattern))
    return hits[0] if hits else None


freq = find("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
mfreq = find("/sys/class/drm/card*/device/hwmon/hwmon*/freq2_input")
power = (find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")
         or find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
busy = find("/sys/class/drm/card*/device/gpu_busy_percent")
print("sysfs:", freq, mfreq, power, busy)


def read(path):
    try:
        with open(path) as fid:
            return float(fid.read().split()[0])
    except (OSError, ValueError, TypeError, IndexError):
        return float("nan")


child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
rows = []
while child.poll() is None:
    rows.append((time.perf_counter(), read(freq) / 1e6, read(mfreq) / 1e6, read(power) / 1e6, read(busy)))
    time.sleep(0.01)
out = child.stdout.read()
print(out.strip().splitlines()[-1][:300] if out.strip() else "(no output)")
active = [r for r in rows if r[4] >= 50] or rows
for name, k in (("shader MHz", 1), ("memory MHz", 2), ("power W", 3)):
    v = [r[k] for r in active if r[k] == r[k]]
    if v:
        print(f"{name}: while busy min {min(v):.0f} median {statistics.median(v):.0f} max {max(v):.0f} "
              f"({len(v)} samples); all samples max {max(r[k] for r in rows if r[k] == r[k]):.0f}")
sys.exit(child.returncode)

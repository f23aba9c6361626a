#!/bin/bash
# A/B of library builds on a bench workload, alternating, REPS rounds of 20 stitches each:
#   tools/ab_libs.sh WORKLOAD REPS NAME1 NAME2 ...     (NAME = base | a build/variants/ name)
export PANO_BENCH_FULL_LINE=1   # the whole record on stdout (bench.py prints a compact line otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=$1; REPS=$2; shift; shift
mkdir -p gpurun_out/ab; LOG=gpurun_out/ab/log_$WL.txt; : > "$LOG"
for r in $(seq "$REPS"); do
  for v in "$@"; do
    if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
    PANO_LIB=$LIB timeout -k 10 300 python bench.py --workload "$WL" --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>gpurun_out/ab/err_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$v', d['ms_per_step'], k.get('blur_mfma_kernel', 0) + k.get('blur_lean_kernel', 0) + k.get('blur_lean5_kernel', 0), 0.0, k.get('multiband_compose_kernel', 0), k.get('warp_windows_kernel', 0), k.get('ownership_cameras_kernel', 0), k.get('owned_boxes_kernel', 0) + k.get('owned_spans_kernel', 0), d.get('ms_per_stitch_one_in_flight') or 0)" >> "$LOG" || echo "$v FAILED" >> "$LOG"
  done
done
python - "$LOG" <<'P'
import collections, statistics, sys
acc = collections.defaultdict(list)
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) == 9:
        acc[f[0]].append([float(x) for x in f[1:]])
    else:
        print(line.strip())
for k, rows in acc.items():
    cols = list(zip(*rows))
    print('%-14s step median %.3f min %.3f | blur %.3f (+ irregular %.3f) compose %.3f warp %.3f own %.3f regions %.3f | one in flight %.3f  (n=%d)' % (
        k, statistics.median(cols[0]), min(cols[0]), statistics.median(cols[1]),
        statistics.median(cols[2]), statistics.median(cols[3]), statistics.median(cols[4]),
        statistics.median(cols[5]), statistics.median(cols[6]), statistics.median(cols[7]), len(rows)))
P

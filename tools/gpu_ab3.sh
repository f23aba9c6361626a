#!/bin/bash
# Robust A/B of an environment switch: REPS alternating rounds, 20 steps each, median per value.
#   tools/gpu_ab3.sh WORKLOAD VAR REPS val1 val2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=$1; VAR=$2; REPS=$3; shift; shift; shift
mkdir -p gpurun_out/ab3; : > gpurun_out/ab3/log.txt
for r in $(seq $REPS); do
  for v in "$@"; do
    env $VAR=$v timeout 300 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])" >> gpurun_out/ab3/log.txt
  done
done
python - <<'P'
import collections, statistics
acc = collections.defaultdict(list)
for line in open('gpurun_out/ab3/log.txt'):
    k, v = line.split(); acc[k].append(float(v))
for k, v in acc.items():
    print(k, 'median %.3f  min %.3f  max %.3f  n=%d' % (statistics.median(v), min(v), max(v), len(v)))
P

#!/bin/bash
# A/B of an environment switch on a bench workload, alternating: tools/ab_env.sh WORKLOAD REPS VAR VALUE_A VALUE_B
export PANO_BENCH_FULL_LINE=1   # the whole record on stdout (bench.py prints a compact line otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=$1; REPS=$2; VAR=$3; shift; shift; shift
for r in $(seq "$REPS"); do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 300 python bench.py --workload "$WL" --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL $VAR=$v', round(d['ms_per_step'],4))"
  done
done

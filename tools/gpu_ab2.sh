#!/bin/bash
# A/B of an environment switch on a bench workload: tools/gpu_ab2.sh WORKLOAD VAR val1 val2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=$1; VAR=$2; shift; shift
for v in "$@"; do
  env $VAR=$v timeout 300 python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL $VAR=$v step %.3f ms' % d['ms_per_step'], {k: round(x, 3) for k, x in d['kernel_ms_per_step'].items() if x > 0.05})"
done

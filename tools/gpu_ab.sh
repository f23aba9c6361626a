#!/bin/bash
# A/B of an environment switch on the cfg3 bench: tools/gpu_ab.sh VAR val1 val2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v step %.3f ms' % d['ms_per_step'], {k: round(x, 3) for k, x in d['kernel_ms_per_step'].items()})"
done

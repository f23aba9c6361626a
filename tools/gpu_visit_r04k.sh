#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04k; mkdir -p $OUT
tools/ab_libs.sh cfg3 3 base ntstore | tee $OUT/ab_ntstore_cfg3.txt
timeout -k 10 300 python tools/host_profile.py cfg2 > $OUT/host_profile_cfg2.txt 2>&1; head -40 $OUT/host_profile_cfg2.txt
for k in 1 2; do
  timeout -k 10 300 python bench.py --workload cfg2 --steps 50 --warmup 5 --in-flight $k --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 in flight $k: ms/step %.3f' % d['ms_per_step'], 'instrumented %.3f' % d['instrumented_ms_per_step'], 'kernels %.3f' % sum(d['kernel_ms_per_step'].values()))"
done

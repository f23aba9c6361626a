#!/bin/bash
# A/B of library builds on the cfg4 --detect bench: tools/ab_detect.sh NAME1 NAME2 ...
export PANO_BENCH_FULL_LINE=1   # the whole record on stdout (bench.py prints a compact line otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
  PANO_LIB=$LIB timeout -k 10 300 python bench.py --workload cfg4 --detect --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms_per_step'].items()})"
done

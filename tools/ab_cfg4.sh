#!/bin/bash
# A/B of library builds on the cfg4 bench (scale space of a 4K frame): tools/ab_cfg4.sh NAME1 NAME2 ...
export PANO_BENCH_FULL_LINE=1   # the whole record on stdout (bench.py prints a compact line otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "$@"; do
  if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
  PANO_LIB=$LIB timeout -k 10 300 python bench.py --workload cfg4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), round(d['roofline']['frac'],3))"
done

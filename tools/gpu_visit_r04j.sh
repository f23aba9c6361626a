#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04j; mkdir -p $OUT
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
tools/ab_libs.sh cfg3 4 base oldlean | tee $OUT/ab_cfg3.txt
tools/ab_libs.sh cfg2 3 base oldlean | tee $OUT/ab_cfg2.txt
tools/ab_libs.sh cfg5 2 base oldlean | tee $OUT/ab_cfg5.txt

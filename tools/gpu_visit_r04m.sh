#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04m; mkdir -p $OUT
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; rc=$?; tail -12 "$OUT/pytest_gpu.log"
[ $rc -ne 0 ] && { echo "TESTS FAILED"; exit 1; }
tools/ab_libs.sh cfg5 2 base nofive | tee $OUT/ab_cfg5.txt
tools/ab_libs.sh cfg3 3 base nofive | tee $OUT/ab_cfg3.txt

#!/bin/bash
# GPU parity tests, then an A/B of library variants on cfg2, cfg3 and cfg5:
#   tools/gpu_visit_ab3.sh <tag> <reps> base VARIANT...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1; REPS=$2; shift; shift
OUT=gpurun_out/$T
mkdir -p "$OUT"
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; rc=$?; tail -15 "$OUT/pytest_gpu.log"
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit 1; }
for WL in cfg2 cfg3 cfg5; do
  echo "== $WL"
  tools/ab_libs.sh "$WL" "$REPS" "$@" | tee "$OUT/ab_$WL.txt"
done

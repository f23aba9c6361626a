#!/bin/bash
# Round 5: a row's two taps as one eight-byte load (warp, the collapse's interior pixels, ownership's
# samplers share the helper) - full tests, A/B against the previous commit
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05v}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
tools/ab_libs.sh cfg3 3 prev base | tee "$OUT/ab_prev_cfg3.txt"
tools/ab_libs.sh cfg5 1 prev base | tee "$OUT/ab_prev_cfg5.txt"
tools/ab_libs.sh cfg2 2 prev base | tee "$OUT/ab_prev_cfg2.txt"
exit 0

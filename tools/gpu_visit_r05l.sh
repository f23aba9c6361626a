#!/bin/bash
# Round 5: the sort kernel's cost again (parallel order statistics), with and without the estimate.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05l}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu (strips, segments, blur)"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "strip or segment or blur or closed_360 or trusted" > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
for t in 0 -1; do
echo "-- PANO_BLUR_SEG_T=$t"
PANO_BLUR_SEG_T=$t tools/gpu_profile.sh "$T" cfg2 30 | grep "mb_sort\|mb_items\|blur_lean\|bench (under"
done
echo "== world-8 strips, three lanes, trusted"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 300 python tools/strip_floor.py cfg3 8 2>/dev/null | grep "^world"
grep -l "GPU core dump" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null && exit 1
exit 0

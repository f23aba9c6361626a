#!/bin/bash
# Round 5: kept geometry (trust_layout = 3): tests, then strip floors with and without it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05z1}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trusted or kept or native or layout" > "$OUT/pytest_kept.log" 2>&1
tail -3 "$OUT/pytest_kept.log"
grep -q " passed" "$OUT/pytest_kept.log" || { tail -60 "$OUT/pytest_kept.log"; exit 1; }
grep -q "failed" "$OUT/pytest_kept.log" && { tail -80 "$OUT/pytest_kept.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
for keep in 0 1; do
  PANO_KEEP_GEOMETRY=$keep PANO_SETS_IN_FLIGHT=3 PANO_PLAN_CACHED=1 timeout -k 10 600 python tools/strip_floor.py cfg3 1 2 4 8 > "$OUT/strip_cfg3_keep$keep.txt" 2>&1 || { tail -30 "$OUT/strip_cfg3_keep$keep.txt"; exit 1; }
  echo "keep=$keep"; grep -E "^world" "$OUT/strip_cfg3_keep$keep.txt" | cut -c1-150
done
PANO_KEEP_GEOMETRY=1 PANO_SETS_IN_FLIGHT=1 PANO_PLAN_CACHED=1 timeout -k 10 600 python tools/strip_floor.py cfg3 1 8 > "$OUT/strip_cfg3_keep1_1lane.txt" 2>&1
echo "keep=1, one lane"; grep -E "^world" "$OUT/strip_cfg3_keep1_1lane.txt" | cut -c1-150
grep -l "GPU core dump" "$OUT"/*.txt 2>/dev/null && exit 1
exit 0

#!/bin/bash
# Round 5, eighth visit: full tests (trusted layouts at full size); the blur's segment length on a
# world-8 strip forced through PANO_BLUR_SEG_T; four lanes with trusted layouts.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05h}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
fault() { if grep -l "GPU core dump\|Memory access fault" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null; then echo "GPU FAULT"; exit 1; fi; return 0; }
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
fault
echo "== world-8 strip: the blur's segment length forced (PANO_BLUR_SEG_T; 0 = the model, -1 = no cut), one lane"
for t in 0 -1 12 16 20 24 32 40 48; do
  PANO_BLUR_SEG_T=$t PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=1 timeout -k 10 300 python tools/strip_floor.py cfg3 8 2>/dev/null | grep "^world" | sed "s/^/T=$t: /"
  fault
done | tee "$OUT/blur_seg_t_strip8.txt"
echo "== four lanes, trusted"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=4 timeout -k 10 300 python tools/strip_floor.py cfg3 8 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_4lanes.txt"
fault

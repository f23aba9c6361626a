#!/bin/bash
# Round 5: the blur's segment length on strips of world 2 / 4 / 8 (and the whole mosaic): forced T
# against the estimate's choice (T=0) - blur kernel ms and ms per stitch, kept geometry, three lanes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05z2}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
export PANO_KEEP_GEOMETRY=1 PANO_SETS_IN_FLIGHT=3 PANO_PLAN_CACHED=1
for wr in "1 0" "2 0" "4 2" "8 4"; do
  set -- $wr
  for t in 0 -1 6 8 12 16 24 32 48 64 96; do
    PANO_BLUR_SEG_T=$t PANO_STRIP_RANK=$2 timeout -k 10 300 python tools/strip_floor.py cfg3 $1 > "$OUT/seg_w$1_t$t.txt" 2>&1 || { tail -20 "$OUT/seg_w$1_t$t.txt"; exit 1; }
    grep -q "GPU core dump" "$OUT/seg_w$1_t$t.txt" && exit 1
    python - "$OUT/seg_w$1_t$t.txt" $1 $t <<'P'
import re, sys
line = [l for l in open(sys.argv[1]) if l.startswith("world")][0]
ms = float(re.search(r": ([0-9.]+) ms per stitch", line).group(1))
blur = float(re.search(r"'blur_lean_kernel': ([0-9.]+)", line).group(1))
print("world %s T %4s: %.3f ms per stitch, blur %.3f" % (sys.argv[2], sys.argv[3], ms, blur))
P
  done
done | tee "$OUT/segment_scan_cfg3.txt"
exit 0

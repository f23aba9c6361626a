#!/bin/bash
# Round 5: lanes per rank with the geometry kept (world 8 strip, config 3)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05z4}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
export PANO_KEEP_GEOMETRY=1 PANO_PLAN_CACHED=1 PANO_STRIP_RANK=4
for lanes in 1 2 3 4 6 3; do
  PANO_SETS_IN_FLIGHT=$lanes timeout -k 10 300 python tools/strip_floor.py cfg3 8 > "$OUT/lanes$lanes.txt" 2>&1 || { tail -20 "$OUT/lanes$lanes.txt"; exit 1; }
  grep -q "GPU core dump" "$OUT/lanes$lanes.txt" && exit 1
  echo "lanes $lanes: $(grep '^world' "$OUT/lanes$lanes.txt" | cut -c1-110)"
done | tee "$OUT/kept_geometry_lanes_world8.txt"
exit 0

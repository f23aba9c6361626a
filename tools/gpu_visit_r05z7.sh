#!/bin/bash
# Round 5: the column-cost model's weights (mosaic, valid, near-seam) on a world-8 / world-4 split of config 3
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05z7}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
for keep in 0 1; do for sh in 0.27,0.25,0.98 0.27,0.25,1.5 0.27,0.25,2.5 0.27,0.25,4 0.15,0.25,1.5 0.15,0.15,2.5 0.4,0.25,0.98; do
  PANO_COST_SHARES=$sh PANO_KEEP_GEOMETRY=$keep PANO_SETS_IN_FLIGHT=3 PANO_PLAN_CACHED=1 timeout -k 10 600 python tools/strip_floor.py cfg3 4 8 > "$OUT/strip_keep${keep}_$sh.txt" 2>&1 || { tail -30 "$OUT/strip_keep${keep}_$sh.txt"; exit 1; }
  grep -q "GPU core dump" "$OUT/strip_keep${keep}_$sh.txt" && exit 1
  echo "== kept geometry $keep, shares $sh"; grep -E "^world" "$OUT/strip_keep${keep}_$sh.txt" | cut -c1-120
done; done | tee "$OUT/cost_shares_scan_cfg3.txt"
exit 0

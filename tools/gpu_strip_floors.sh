#!/bin/bash
# Strip floors (tools/strip_floor.py) of config 3 and config 5, one and two stitches in flight, plan per
# stitch and cached:   tools/gpu_strip_floors.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-floors}; mkdir -p $OUT
for inflight in 1 2; do
  for cached in 0 1; do
    sfx=""; [ $inflight = 2 ] && sfx="_two_in_flight"; [ $cached = 1 ] && sfx="${sfx}_plan_cached"
    PANO_SETS_IN_FLIGHT=$inflight PANO_PLAN_CACHED=$cached timeout -k 10 500 python tools/strip_floor.py cfg3 1 2 4 8 > $OUT/strip_floor_cfg3$sfx.txt 2>&1
    grep "^world" $OUT/strip_floor_cfg3$sfx.txt | cut -c1-110
    PANO_DISTINCT_FRAMES=6 PANO_SETS_IN_FLIGHT=$inflight PANO_PLAN_CACHED=$cached timeout -k 10 500 python tools/strip_floor.py cfg5 1 2 4 8 > $OUT/strip_floor_cfg5$sfx.txt 2>&1
    grep "^world" $OUT/strip_floor_cfg5$sfx.txt | cut -c1-110
  done
done

#!/bin/bash
# Round 5: is a world-8 strip bound by the host or by the GPU? (queueing time against wall time, cProfile)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05x}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
PANO_SETS_IN_FLIGHT=3 PANO_PLAN_CACHED=1 PANO_HOST_PROFILE=1 timeout -k 10 600 python tools/strip_floor.py cfg3 8 > "$OUT/strip_host_3lanes.txt" 2>&1 || { tail -30 "$OUT/strip_host_3lanes.txt"; exit 1; }
grep -E "^world|Plan alone" "$OUT/strip_host_3lanes.txt"
PANO_SETS_IN_FLIGHT=1 PANO_PLAN_CACHED=1 timeout -k 10 600 python tools/strip_floor.py cfg3 8 > "$OUT/strip_host_1lane.txt" 2>&1 || { tail -30 "$OUT/strip_host_1lane.txt"; exit 1; }
grep -E "^world" "$OUT/strip_host_1lane.txt"
grep -l "GPU core dump" "$OUT"/*.txt 2>/dev/null && exit 1
exit 0

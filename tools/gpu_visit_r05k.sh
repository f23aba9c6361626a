#!/bin/bash
# Round 5: the sort kernel's cost after the list-schedule estimate was made cheap (items in
# registers, eight bins per LDS read, no estimate for launches of three rounds and more).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05k}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu (strips, segments, blur)"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "strip or segment or blur or full_size or closed_360 or trusted or rccl" > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
for wl in cfg2 cfg3; do tools/gpu_profile.sh "$T" $wl 30 | grep "mb_sort\|mb_items\|blur_lean\|bench (under"; done
echo "== a world-8 strip under the profiler"
HERE=$PWD; cd /tmp
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$HERE/$OUT/trace_strip" -- python3 "$HERE/tools/strip_floor.py" cfg3 8 > "$HERE/$OUT/strip_profiled.txt" 2>/dev/null
cd "$HERE"
grep "^world" "$OUT/strip_profiled.txt"
s=$(find "$OUT/trace_strip" -name "*kernel_stats.csv" | head -1)
python3 - "$s" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in ("mb_sort", "mb_items", "blur_lean", "ownership", "layout", "tile_flags", "interior", "block_owner", "owned_spans", "init_regions", "warp", "compose")):
        print("%-40s calls %6s avg %8.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
P
find "$OUT/trace_strip" -name "*kernel_trace.csv" -delete
echo "== world-8 strips, three lanes, trusted"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 300 python tools/strip_floor.py cfg3 1 8 2>/dev/null | grep "^world"
grep -l "GPU core dump" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null && exit 1
exit 0

#!/bin/bash
# Round 5: how the kernels of three strip stitches in flight share the GPU (world 8, rank 4):
# kernel trace of tools/strip_floor.py, tools/trace_overlap.py over its 20 timed stitches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HERE=$PWD
T=${1:-r05y}
OUT=$HERE/gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
export PANO_SETS_IN_FLIGHT=3 PANO_PLAN_CACHED=1 PANO_STRIP_RANK=4
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- \
    python3 "$HERE/tools/strip_floor.py" cfg3 8 > "$OUT/strip.txt" 2> "$OUT/rocprof.log"
echo "rocprof rc=$?"
cd "$HERE"
grep -E "^world" "$OUT/strip.txt"
grep -l "GPU core dump" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null && exit 1
t=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_overlap.py "$t" --steps 20 | tee "$OUT/overlap_strip_world8.txt"
rm -rf "$OUT/trace"
exit 0

#!/bin/bash
# Kernel trace of the bench with the trace kept for tools/trace_overlap.py:  tools/gpu_overlap.sh <tag> [workload] [steps] [bench args...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HERE=$PWD; OUT=$HERE/gpurun_out/${1:-ovl}; WL=${2:-cfg3}; STEPS=${3:-50}; shift; shift; shift
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$WL" -- \
    python3 "$HERE/bench.py" --workload "$WL" --steps "$STEPS" --warmup 5 --no-cpu-baseline --no-secondary --busy-seconds 0 "$@" \
    > "$OUT/bench_${WL}.json" 2> "$OUT/rocprof_$WL.log"
echo "rocprof rc=$?"
cd "$HERE"
t=$(find "$OUT/trace_$WL" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_overlap.py "$t" --steps "$STEPS" | tee "$OUT/overlap_$WL.txt"
rm -f "$t"

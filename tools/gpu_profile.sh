#!/bin/bash
# Steady-state profile of one bench workload: rocprofv3 kernel trace of
#   python3 bench.py --workload WL --steps 50 --warmup 5 --no-cpu-baseline --no-secondary
# reduced by tools/trace_stats.py (first dispatches dropped; the instrumented pass apart), with the
# bench line of the SAME run beside it.   tools/gpu_profile.sh <tag> [workload] [steps]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HERE=$PWD
OUT=$HERE/gpurun_out/${1:-prof}
WL=${2:-cfg3}
STEPS=${3:-50}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$WL" -- \
    python3 "$HERE/bench.py" --workload "$WL" --steps "$STEPS" --warmup 5 --no-cpu-baseline --no-secondary --busy-seconds 0 --side-file "$OUT/bench_${WL}_profiled_full.json" ${BENCH_EXTRA:-} \
    > "$OUT/bench_${WL}_profiled.json" 2> "$OUT/rocprof_$WL.log"
echo "rocprof rc=$?"
cd "$HERE"
t=$(find "$OUT/trace_$WL" -name "*kernel_trace.csv" | head -1)
s=$(find "$OUT/trace_$WL" -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp "$s" "$OUT/${WL}_kernel_stats_raw.csv"
if [ -n "$t" ]; then
  python3 tools/trace_stats.py "$t" --steps "$STEPS" --out "$OUT/${WL}_kernel_stats_steady.csv" | tee "$OUT/${WL}_kernel_stats_steady.txt"
  rm -f "$t"
fi
python3 - "$OUT/bench_${WL}_profiled.json" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("bench (under the profiler): ms/step %.3f; roofline %s avg launch %.4f ms frac %.3f" % (
        d["ms_per_step"], r["kernel"], r.get("avg_launch_ms", 0), r["frac"]))
except Exception as e:
    print("no bench line:", e)
P

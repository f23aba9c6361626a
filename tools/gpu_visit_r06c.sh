#!/bin/bash
# Round 6: the native detect call with graph replay - its parity test, the SIFT tests, the jittered
# full-size rig, then config 4 with and without detection (bench lines), graphs on and off.
#   tools/gpu_visit_r06c.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06c}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest (sift, jittered rig)"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "sift or jittered or scale_space" > "$OUT/pytest_sift.log" 2>&1; rc=$?; tail -4 "$OUT/pytest_sift.log"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest_sift.log" | head -20; exit 1; }
for mode in graph nograph; do
  for det in "" "--detect"; do
    tag="cfg4${det:+_detect}_$mode"
    if [ $mode = nograph ]; then export PANO_SIFT_GRAPH=0; else unset PANO_SIFT_GRAPH; fi
    timeout -k 10 600 python bench.py --workload cfg4 $det --steps 30 --warmup 4 --no-cpu-baseline --side-file "$OUT/bench_${tag}_full.json" > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err" || { tail -5 "$OUT/bench_$tag.err"; exit 1; }
    python - "$OUT/bench_$tag.json" "$tag" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], d.get("settings"), "kp", d["config"].get("keypoints_per_frame"))
P
  done
done
unset PANO_SIFT_GRAPH
echo "== kernel trace of the detect path (which kernels the sort runs)"
BENCH_EXTRA=--detect tools/gpu_profile.sh "$T" cfg4 30

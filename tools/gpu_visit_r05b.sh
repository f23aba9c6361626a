#!/bin/bash
# Round 5, second visit: the two-level ownership kernel - GPU tests, A/B against round 4's kernel
# (PANO_OWN_PRUNE=3) on configs 3 / 5 / 2 and on a world-8 strip - and the hardware-queue question
# (GPU_MAX_HW_QUEUES) for the secondaries that run two lanes late in the default run.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05b}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -6 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || exit 1
grep -q "failed" "$OUT/pytest_gpu.log" && exit 1
ab() {   # workload reps
  for r in $(seq "$2"); do
    for v in 3 1; do
      PANO_OWN_PRUNE=$v timeout -k 10 300 python bench.py --workload "$1" --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --busy-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1 own_prune=$v step %.4f one-in-flight %s ownership %.4f regions %.4f' % (d['ms_per_step'], d.get('ms_per_stitch_one_in_flight'), k.get('ownership_cameras_kernel', 0), k.get('owned_boxes_kernel', 0) + k.get('owned_spans_kernel', 0)))"
    done
  done
}
echo "== ownership A/B (3 = round 4's one-level kernel, 1 = two levels)"
ab cfg3 3 | tee "$OUT/ab_ownership_cfg3.txt"
ab cfg2 2 | tee "$OUT/ab_ownership_cfg2.txt"
PANO_DISTINCT_FRAMES=6 ab cfg5 1 | tee "$OUT/ab_ownership_cfg5.txt"
echo "== world-8 strip, two lanes, plan from the memo"
for v in 3 1; do
  PANO_OWN_PRUNE=$v PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 300 python tools/strip_floor.py cfg3 8 2>/dev/null | grep "^world" | sed "s/^/own_prune=$v: /"
done | tee "$OUT/ab_ownership_strip8.txt"
echo "== the default run with 8 hardware queues (cfg2 / cfg4_detect with two lanes late in the run)"
for q in 8 4; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --busy-seconds 0 > "$OUT/bench_default_hwq$q.json" 2>/dev/null
python - "$OUT/bench_default_hwq$q.json" $q <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES", sys.argv[2], "cfg3 ms/step %.3f one-in-flight %.3f" % (d["ms_per_step"], d["ms_per_stitch_one_in_flight"]))
for k, v in d.get("secondary", {}).items():
    print("   ", k, "ms/step", v.get("ms_per_step"), "instr", v.get("instrumented_ms_per_step"), v.get("error", ""))
P
done

#!/bin/bash
# Round 5, first visit: the GPU tests (with the RCCL world-1 test and the float criterion at full
# size), strip floors of config 3 with 1 - 4 lanes per rank and the plan out of the memo, the
# driver's command, config 4's detection alone.      tools/gpu_visit_r05a.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05a}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>&1 | head -12; nproc; } > "$OUT/info.log"
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > "$OUT/pytest_gpu.log" 2>&1; tail -18 "$OUT/pytest_gpu.log"
echo "== strip floors, config 3, world 8, plan from the memo, lanes 1..4"
for l in 1 2 3 4; do
  PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=$l timeout -k 10 300 python tools/strip_floor.py cfg3 8 --json=$OUT/strip_floor_cfg3_lanes.json 2>/dev/null | grep -v amdgpu.ids | sed "s/^/lanes $l: /"
done
echo "== the same with the record layout on the device (PANO_STITCH_ASYNC=1)"
for l in 1 2 3; do
  PANO_STITCH_ASYNC=1 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=$l timeout -k 10 300 python tools/strip_floor.py cfg3 8 --json=$OUT/strip_floor_cfg3_lanes_async.json 2>/dev/null | grep -v amdgpu.ids | sed "s/^/async lanes $l: /"
done
echo "== world 1 2 4 8, two lanes, plan from the memo"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 400 python tools/strip_floor.py cfg3 1 2 4 8 --json=$OUT/strip_floor_cfg3.json 2>/dev/null | grep -v amdgpu.ids
echo "== the driver's command"
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("cfg3 ms/step %.3f value %.0f" % (d["ms_per_step"], d["value"]), r["kernel"], "avg launch %.4f frac %.3f blend_frac %.3f" % (r["avg_launch_ms"], r["frac"], r.get("blend_frac", 0)))
print({k: d.get(k) for k in ("value_kind", "processed_MPps", "pipelined", "ms_per_stitch_one_in_flight", "arithmetic", "ms_per_step_strict_f32", "duty_cycle")})
print("kernels", {k: round(v, 4) for k, v in d["kernel_ms_per_step"].items()})
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("busy_loop"))
for k, v in d.get("secondary", {}).items():
    print(" ", k, "ms/step", v.get("ms_per_step"), "instr", v.get("instrumented_ms_per_step"), "frac", (v.get("roofline") or {}).get("frac"), v.get("error", ""))
P
echo "== config 4 with detection, alone: two frames in flight / one"
for st in 2 1; do
  PANO_CFG4_STREAMS=$st timeout -k 10 300 python bench.py --workload cfg4 --detect --steps 16 --warmup 4 --no-cpu-baseline > "$OUT/bench_cfg4_detect_streams$st.json" 2>/dev/null
  python - "$OUT/bench_cfg4_detect_streams$st.json" $st <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg4 detect, streams", sys.argv[2], "ms/step %.3f instrumented %.3f" % (d["ms_per_step"], d["instrumented_ms_per_step"]), {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()})
P
done

#!/bin/bash
# Round 5, seventh visit: trusted stitches (no wait inside a stitch that repeats the verified
# layout) - tests, strip floors with 1 - 3 lanes, config 2 / 3 with the plan cached; the collapse
# with line-aligned reads of the blurred copies (cheap form).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05g}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
fault() { if grep -l "GPU core dump\|Memory access fault" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null; then echo "GPU FAULT"; exit 1; fi; return 0; }
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
fault
echo "== strip floors, config 3, plan from the memo: trusted layouts on / off, lanes 1..3"
for tr in 1 0; do for l in 1 2 3; do
  PANO_TRUST_LAYOUT=$tr PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=$l timeout -k 10 300 python tools/strip_floor.py cfg3 8 --json=$OUT/strip_floor_cfg3_trust.json 2>/dev/null | grep "^world" | sed "s/^/trust $tr lanes $l: /"
done; done | tee "$OUT/strip_floor_trust.txt"
fault
echo "== world 1 2 4 8, two lanes, trusted"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 400 python tools/strip_floor.py cfg3 1 2 4 8 --json=$OUT/strip_floor_cfg3.json 2>/dev/null | grep -v amdgpu.ids | tee "$OUT/strip_floor_cfg3.txt"
fault
echo "== the driver's command"
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("cfg3 ms/step %.3f value %.0f" % (d["ms_per_step"], d["value"]), r["kernel"], "avg launch %.4f frac %.3f blend_frac %.3f" % (r["avg_launch_ms"], r["frac"], r.get("blend_frac", 0)))
print({k: d.get(k) for k in ("value_kind", "processed_MPps", "pipelined", "ms_per_stitch_one_in_flight", "ms_per_step_strict_f32", "duty_cycle")})
print("kernels", {k: round(v, 4) for k, v in d["kernel_ms_per_step"].items()})
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("busy_loop"))
for k, v in d.get("secondary", {}).items():
    print(" ", k, "ms/step", v.get("ms_per_step"), "instr", v.get("instrumented_ms_per_step"), "frac", (v.get("roofline") or {}).get("frac"), v.get("error", ""))
P
fault
echo "== collapse with line-aligned reads of the blurred copies (timing only)"
tools/ab_libs.sh cfg3 3 base compose_aligned | tee "$OUT/ab_compose_aligned2_cfg3.txt"
fault

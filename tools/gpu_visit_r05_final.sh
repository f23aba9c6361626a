#!/bin/bash
# Round 5's closing visit: parity tests, smoke, the driver's command, steady-state profiles,
# counter passes.   tools/gpu_visit_final.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05z}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>&1 | head -12; nproc; } > "$OUT/info.log"
echo "== pytest -m gpu"
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
echo "== the driver's command"
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("cfg3 ms/step %.3f value %.0f" % (d["ms_per_step"], d["value"]), r["kernel"], "avg launch %.4f frac %.3f blend_frac %.3f" % (r["avg_launch_ms"], r["frac"], r.get("blend_frac", 0)))
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("busy_loop"))
for k, v in d.get("secondary", {}).items():
    print(" ", k, "ms/step", v.get("ms_per_step"), "frac", (v.get("roofline") or {}).get("frac"), v.get("error", ""))
P
echo "== steady-state profiles"
tools/gpu_profile.sh "$T" cfg3 50
tools/gpu_profile.sh "$T" cfg2 50
tools/gpu_profile.sh "$T" cfg5 6
tools/gpu_profile.sh "$T" cfg4 30
echo "== counter passes"
for wl in cfg3 cfg2 cfg5 cfg4; do
  tools/pmc.sh "$T/pmc_$wl" $wl > "$OUT/pmc_$wl.log" 2>&1; tail -1 "$OUT/pmc_$wl.log"
done
echo "== 2-rank dry run on one GPU through bench.py's own launcher (gloo)"
PANO_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-secondary > "$OUT/bench_2rank_selflaunch.json" 2> "$OUT/bench_2rank.err"; cut -c1-700 "$OUT/bench_2rank_selflaunch.json"
echo "== strip floors: config 3 (three lanes, plan memo, trusted layouts, strips of equal work, every rank), the same with the geometry kept (appended to the same file, flagged), equal widths for comparison; config 5 at world 8"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 400 python tools/strip_floor.py cfg3 1 2 4 8 --json=$OUT/strip_floor_cfg3_final.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg3_final.txt"
PANO_KEEP_GEOMETRY=1 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 400 python tools/strip_floor.py cfg3 1 2 4 8 --json=$OUT/strip_floor_cfg3_final.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg3_kept_final.txt"
PANO_STRIP_BALANCE=0 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 400 python tools/strip_floor.py cfg3 1 8 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg3_equal_width.txt"
PANO_DISTINCT_FRAMES=6 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 900 python tools/strip_floor.py cfg5 1 8 --json=$OUT/strip_floor_cfg5_final.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg5_final.txt"
PANO_KEEP_GEOMETRY=1 PANO_DISTINCT_FRAMES=6 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 900 python tools/strip_floor.py cfg5 1 8 --json=$OUT/strip_floor_cfg5_final.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg5_kept_final.txt"
if grep -l "GPU core dump\|Memory access fault" "$OUT"/*.txt "$OUT"/*.log "$OUT"/*.err 2>/dev/null; then echo "GPU FAULT"; exit 1; fi
exit 0

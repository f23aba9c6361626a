#!/bin/bash
# Round 5: trusted stitches with the spans and the cleanup inside the layout kernel (two launches
# less), the sort kernel's candidates by wave - full tests, strips, config 2 / 3 with the plan cached.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05r}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
echo "== strips: one lane (the chain) and three lanes"
for l in 1 3; do
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=$l timeout -k 10 300 python tools/strip_floor.py cfg3 1 4 8 2>/dev/null | grep "^world" | sed "s/^/lanes $l: /" | sed "s/(timed.*//"
done | tee "$OUT/strips.txt"
grep -l "GPU core dump" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null && exit 1
echo "== the driver's command"
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("cfg3 ms/step %.3f value %.0f" % (d["ms_per_step"], d["value"]), r["kernel"], "avg launch %.4f frac %.3f blend_frac %.3f" % (r["avg_launch_ms"], r["frac"], r.get("blend_frac", 0)))
for k, v in d.get("secondary", {}).items():
    print(" ", k, "ms/step", v.get("ms_per_step"), "instr", v.get("instrumented_ms_per_step"), v.get("error", ""))
P
tools/gpu_profile.sh "$T" cfg2 30 | grep "mb_sort\|owned_spans\|init_regions\|layout\|bench (under"
grep -l "GPU core dump" "$OUT"/*.txt "$OUT"/*.log "$OUT"/*.err 2>/dev/null && exit 1
exit 0

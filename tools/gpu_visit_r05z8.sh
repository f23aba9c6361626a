#!/bin/bash
# Round 5: the strip floors of the closing visit again, with max(2, lanes) stitches in flight (the oldest is
# waited for before another is queued), and the plan-cached / geometry-kept secondaries of the bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05z8}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 400 python tools/strip_floor.py cfg3 1 2 4 8 --json=$OUT/strip_floor_cfg3_final.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg3_final.txt"
PANO_KEEP_GEOMETRY=1 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 400 python tools/strip_floor.py cfg3 1 2 4 8 --json=$OUT/strip_floor_cfg3_final.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg3_kept_final.txt"
PANO_STRIP_BALANCE=0 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 400 python tools/strip_floor.py cfg3 1 8 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg3_equal_width.txt"
PANO_DISTINCT_FRAMES=6 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 900 python tools/strip_floor.py cfg5 1 8 --json=$OUT/strip_floor_cfg5_final.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg5_final.txt"
PANO_KEEP_GEOMETRY=1 PANO_DISTINCT_FRAMES=6 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 900 python tools/strip_floor.py cfg5 1 8 --json=$OUT/strip_floor_cfg5_final.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_cfg5_kept_final.txt"
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --busy-seconds 0 > "$OUT/bench_secondaries.json" 2> "$OUT/bench_secondaries.err" || { tail -20 "$OUT/bench_secondaries.err"; exit 1; }
python - "$OUT/bench_secondaries.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg3 ms/step %.3f" % d["ms_per_step"])
for k, v in d.get("secondary", {}).items():
    print(" ", k, "ms/step", v.get("ms_per_step"), v.get("error", ""))
P
if grep -l "GPU core dump\|Memory access fault" "$OUT"/*.txt "$OUT"/*.err 2>/dev/null; then echo "GPU FAULT"; exit 1; fi
exit 0

#!/bin/bash
# Round 5: all GPU tests on the kept-geometry code, the driver's bench command with its secondaries
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05z3}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -80 "$OUT/pytest_gpu.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { tail -20 "$OUT/bench_default.err"; exit 1; }
python - "$OUT/bench_default.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("cfg3 ms/step %.3f value %.0f" % (d["ms_per_step"], d["value"]), r["kernel"], "avg launch %.4f frac %.3f" % (r["avg_launch_ms"], r["frac"]))
for k, v in d.get("secondary", {}).items():
    print(" ", k, "ms/step", v.get("ms_per_step"), sorted((v.get("kernel_ms_per_step") or {}).keys())[:12] if "kept" in k else "", v.get("error", ""))
print(json.dumps(d.get("scaling_projection"))[:900])
P
grep -l "GPU core dump" "$OUT"/*.err 2>/dev/null && exit 1
exit 0

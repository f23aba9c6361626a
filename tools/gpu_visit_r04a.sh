#!/bin/bash
# Round 4, first GPU visit: the state the round starts from, measured the way VERDICT r03 item 1 asks.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r04a}
mkdir -p "$OUT"
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>&1 | head -12; nproc; } > "$OUT/info.log"
echo "== pytest -m gpu"
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
echo "== FETCH_SIZE / WRITE_SIZE calibration"
tools/fetch_calib.sh "${1:-r04a}/calib"
echo "== steady-state profile cfg3"
tools/gpu_profile.sh "${1:-r04a}" cfg3 50
echo "== bench cfg3 (no profiler), cfg2"
timeout -k 10 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err"
timeout -k 10 600 python bench.py --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"
python - "$OUT" <<'P'
import json, sys
for wl in ("cfg3", "cfg2"):
    d = json.loads(open(f"{sys.argv[1]}/bench_{wl}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(wl, "ms/step %.3f" % d["ms_per_step"], r["kernel"], "avg launch %.4f ms frac %.3f" % (r["avg_launch_ms"], r["frac"]),
          {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0.01})
P
echo "== PMC passes cfg2"
tools/pmc.sh "${1:-r04a}/pmc_cfg2" cfg2 > "$OUT/pmc_cfg2.log" 2>&1; tail -3 "$OUT/pmc_cfg2.log"

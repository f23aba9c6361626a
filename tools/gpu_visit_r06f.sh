#!/bin/bash
# Round 6: the whole GPU suite (timing), smoke, the driver's command.   tools/gpu_visit_r06f.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06f}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > "$OUT/pytest_gpu.log" 2>&1; rc=$?; tail -18 "$OUT/pytest_gpu.log"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest_gpu.log" | head -30; exit 1; }
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
echo "== the driver's command"
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --side-file "$OUT/bench_default_full.json" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { tail "$OUT/bench_default.err"; exit 1; }
wc -c "$OUT/bench_default.json"; cat "$OUT/bench_default.json"

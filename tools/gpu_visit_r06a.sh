#!/bin/bash
# Round 6, first visit: GPU tests on the tree with the dead blur generations removed, the smoke, the
# driver's command (is the stdout line compact?), a steady-state profile of config 3, and the 2- and
# 4-rank gloo dry runs with the pre-flight.   tools/gpu_visit_r06a.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06a}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>&1 | head -12; nproc; } > "$OUT/info.log"
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; rc=$?; tail -3 "$OUT/pytest_gpu.log"
[ $rc -ne 0 ] && { echo "GPU tests failed: stopping"; exit 1; }
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
echo "== the driver's command"
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --side-file "$OUT/bench_default_full.json" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || exit 1
wc -c "$OUT/bench_default.json"
cat "$OUT/bench_default.json"
echo "== steady-state profile"
tools/gpu_profile.sh "$T" cfg3 50 || exit 1
echo "== 2-rank and 4-rank dry runs on one GPU through bench.py's own launcher (gloo)"
for n in 2 4; do
  PANO_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus $n --steps 3 --warmup 1 --side-file "$OUT/bench_${n}rank_full.json" > "$OUT/bench_${n}rank_selflaunch.json" 2> "$OUT/bench_${n}rank.err" || { tail -5 "$OUT/bench_${n}rank.err"; exit 1; }
  wc -c "$OUT/bench_${n}rank_selflaunch.json"; cut -c1-1500 "$OUT/bench_${n}rank_selflaunch.json"; grep "bench preflight" "$OUT/bench_${n}rank.err"
done

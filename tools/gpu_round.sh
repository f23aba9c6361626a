#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench lines, rocprofv3 kernel stats.
# Everything lands in gpurun_out/<tag>/ (merged back into the repo by gpurun).
#   tools/gpu_round.sh <tag> [quick]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r01}
MODE=${2:-full}
mkdir -p "$OUT"
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>&1 | head -12; nproc; } > "$OUT/info.log"
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee "$OUT/pytest_gpu.log"
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log"
echo "== bench cfg2"
timeout 600 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_cfg2.json"
echo "== bench cfg3"
if [ "$MODE" = quick ]; then
  timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee "$OUT/bench_cfg3.json"
else
  timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 | tee "$OUT/bench_cfg3.json"
  echo "== 2-rank dry run on one GPU (gloo rendezvous, shared device)"
  PANO_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | tail -2 | tee "$OUT/bench_cfg3_2rank_dryrun.json"
fi
echo "== rocprof"
HERE=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$HERE/$OUT/prof_cfg3" -- python3 "$HERE/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$HERE/$OUT/rocprof.log" 2>&1
cd "$HERE"
f=$(find "$OUT/prof_cfg3" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f" | cut -c1-160
find "$OUT/prof_cfg3" -name "*kernel_trace.csv" -size +20M -delete

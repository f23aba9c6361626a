#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench lines, rocprofv3 kernel stats.
# Everything lands in gpurun_out/<tag>/ (merged back into the repo by gpurun).
#   tools/gpu_round.sh <tag> [quick]
set -u
export PANO_BENCH_FULL_LINE=1   # the whole record on stdout (bench.py prints a compact line otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r03}
MODE=${2:-full}
mkdir -p "$OUT"
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>&1 | head -12; nproc; } > "$OUT/info.log"
echo "== pytest -m gpu"
timeout -k 10 1500 python -m pytest tests -m gpu -x -q -s > "$OUT/pytest_gpu.log" 2>&1; tail -5 "$OUT/pytest_gpu.log"
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log"
for wl in cfg2 cfg3 cfg4; do
  echo "== bench $wl"
  extra="--no-cpu-baseline --no-secondary"; [ "$MODE" = full ] && extra="--no-secondary"
  timeout -k 10 900 python bench.py --workload $wl --steps 20 --warmup 3 $extra 2> "$OUT/bench_$wl.err" | tail -1 > "$OUT/bench_$wl.json"
  python - "$OUT/bench_$wl.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step %.3f value %.0f %s" % (d["ms_per_step"], d["value"], d["unit"]), "roofline", d["roofline"]["kernel"], "%.3f" % d["roofline"]["frac"],
      {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items() if v > 0.01})
P
done
if [ "$MODE" = full ]; then
  echo "== bench cfg4 with keypoints and descriptors"
  timeout -k 10 900 python bench.py --workload cfg4 --detect --steps 8 --warmup 2 --no-cpu-baseline 2> "$OUT/bench_cfg4_detect.err" | tail -1 > "$OUT/bench_cfg4_detect.json"; cut -c1-200 "$OUT/bench_cfg4_detect.json"
  echo "== bench cfg5 (one GPU)"
  timeout -k 10 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2> "$OUT/bench_cfg5.err" | tail -1 > "$OUT/bench_cfg5.json"; cut -c1-400 "$OUT/bench_cfg5.json"
  echo "== 2-rank dry run on one GPU (gloo rendezvous, shared device)"
  PANO_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
      --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 2> "$OUT/bench_2rank.err" | tail -1 > "$OUT/bench_cfg3_2rank_dryrun.json"; cut -c1-600 "$OUT/bench_cfg3_2rank_dryrun.json"
  echo "== rocprof"
  HERE=$PWD
  cd /tmp
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$HERE/$OUT/prof_cfg3" -- python3 "$HERE/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$HERE/$OUT/rocprof.log" 2>&1
  cd "$HERE"
  f=$(find "$OUT/prof_cfg3" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f"
  find "$OUT/prof_cfg3" -name "*kernel_trace.csv" -size +20M -delete
  for wl in cfg4 cfg5; do
    cd /tmp
    timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$HERE/$OUT/prof_$wl" -- python3 "$HERE/bench.py" --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > "$HERE/$OUT/rocprof_$wl.log" 2>&1
    cd "$HERE"
    f=$(find "$OUT/prof_$wl" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-160
    find "$OUT/prof_$wl" -name "*kernel_trace.csv" -delete
  done
fi

#!/bin/bash
# Round 6's closing visit, in two calls (a call is at most 20 minutes):
#   tools/gpu_visit_r06_final.sh <tag> a   GPU tests, smoke, the driver's command, steady-state profiles
#                                          (config 3 / 2 / 5 / 4 / 4 with detection), 2- and 4-rank dry runs
#   tools/gpu_visit_r06_final.sh <tag> b   counter passes of the four workloads, the ownership kernel's issue share
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06z}
PART=${2:-a}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$PART" = a ]; then
  { rocm-smi --showproductname 2>&1 | head -12; nproc; } > "$OUT/info.log"
  echo "== pytest -m gpu"
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; rc=$?; tail -3 "$OUT/pytest_gpu.log"
  [ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest_gpu.log" | head -30; exit 1; }
  echo "== smoke"
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
  echo "== the driver's command"
  timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --side-file "$OUT/bench_default_full.json" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { tail "$OUT/bench_default.err"; exit 1; }
  wc -c "$OUT/bench_default.json"; cat "$OUT/bench_default.json"
  echo "== steady-state profiles"
  tools/gpu_profile.sh "$T" cfg3 50 || exit 1
  tools/gpu_profile.sh "$T" cfg2 50 || exit 1
  tools/gpu_profile.sh "$T" cfg5 6 || exit 1
  tools/gpu_profile.sh "$T" cfg4 30 || exit 1
  for f in bench_cfg4_profiled.json bench_cfg4_profiled_full.json cfg4_kernel_stats_steady.txt cfg4_kernel_stats_steady.csv cfg4_kernel_stats_raw.csv; do mv "$OUT/$f" "$OUT/${f/cfg4/cfg4_scale_space}"; done
  BENCH_EXTRA=--detect tools/gpu_profile.sh "$T" cfg4 30 || exit 1
  for f in bench_cfg4_profiled.json bench_cfg4_profiled_full.json cfg4_kernel_stats_steady.txt cfg4_kernel_stats_steady.csv cfg4_kernel_stats_raw.csv; do mv "$OUT/$f" "$OUT/${f/cfg4/cfg4_detect}"; done
  echo "== 2- and 4-rank dry runs on one GPU through bench.py's own launcher (gloo)"
  for n in 2 4; do
    PANO_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus $n --steps 3 --warmup 1 --side-file "$OUT/bench_${n}rank_full.json" > "$OUT/bench_${n}rank_selflaunch.json" 2> "$OUT/bench_${n}rank.err" || { tail -5 "$OUT/bench_${n}rank.err"; exit 1; }
    wc -c "$OUT/bench_${n}rank_selflaunch.json"; grep "bench preflight" "$OUT/bench_${n}rank.err" | tee "$OUT/bench_${n}rank_preflight.txt"
  done
else
  echo "== counter passes"
  for wl in cfg3 cfg2 cfg5 cfg4; do
    tools/pmc.sh "$T/pmc_$wl" $wl > "$OUT/pmc_$wl.log" 2>&1; tail -1 "$OUT/pmc_$wl.log"
    cp "$OUT/pmc_$wl/summary.txt" "$OUT/pmc_${wl}_summary.txt"
    cp "$OUT/pmc_$wl/pmc_traffic.json" "$OUT/pmc_traffic_$wl.json"
  done
  for wl in cfg3 cfg2 cfg5; do
    python3 tools/own_issue.py "$OUT/pmc_${wl}_summary.txt" $wl "$OUT/own_issue_$wl.json"
  done
fi

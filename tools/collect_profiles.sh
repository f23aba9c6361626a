#!/bin/bash
# Copies what a tools/gpu_round.sh + tools/pmc.sh visit left under gpurun_out/ into profiles/<round>/
#   tools/collect_profiles.sh <round tag> <pmc tag> <dest, e.g. profiles/r03>
set -u
cd "$(dirname "$0")/.."
R=gpurun_out/$1; P=gpurun_out/$2; D=$3
mkdir -p "$D"
for f in bench_cfg2 bench_cfg3 bench_cfg4 bench_cfg4_detect bench_cfg5 bench_cfg3_2rank_dryrun; do
  [ -s "$R/$f.json" ] && cp "$R/$f.json" "$D/$f.json"
done
for wl in cfg3 cfg4 cfg5; do
  f=$(find "$R/prof_$wl" -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$D/${wl}_kernel_stats.csv"
done
[ -s "$R/pytest_gpu.log" ] && tail -5 "$R/pytest_gpu.log" > "$D/pytest_gpu_tail.txt"
[ -s "$R/smoke.log" ] && cat "$R/smoke.log" >> "$D/pytest_gpu_tail.txt"
[ -s "$P/summary.txt" ] && cp "$P/summary.txt" "$D/pmc_cfg3_summary.txt"
[ -s "$P/pmc_traffic.json" ] && cp "$P/pmc_traffic.json" "$D/pmc_traffic_cfg3.json"
ls "$D" | wc -l

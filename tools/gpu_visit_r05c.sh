#!/bin/bash
# Round 5, third visit: GPU tests of the working tree (warp with SDWA byte offsets and scalar-base
# addressing, per-item blur segments), the ownership kernel's phase timers, A/B against the
# round's first commit (build/variants/r05head).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05c}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -6 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || exit 1
grep -q "failed" "$OUT/pytest_gpu.log" && exit 1
echo "== ownership: phase timers / no evaluation"
for c in cfg3 cfg5; do
PANO_LIB=$PWD/build/variants/ow_stamp/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>/dev/null | tee -a "$OUT/own_stamps.txt"
PANO_LIB=$PWD/build/variants/ow_noeval/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>/dev/null | sed "s/^/no evaluation: /" | tee -a "$OUT/own_stamps.txt"
PANO_OWN_PRUNE=3 timeout -k 10 200 python tools/probe_own_stamps.py $c 2>/dev/null | sed "s/^/round 4 kernel: /" | tee -a "$OUT/own_stamps.txt"
done
echo "== A/B against the round's first commit"
tools/ab_libs.sh cfg3 3 r05head base | tee "$OUT/ab_head_cfg3.txt"
tools/ab_libs.sh cfg2 3 r05head base | tee "$OUT/ab_head_cfg2.txt"
echo "== world-8 strip, two lanes, plan from the memo"
for v in r05head base; do
  if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
  PANO_LIB=$LIB PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 300 python tools/strip_floor.py cfg3 4 8 2>/dev/null | grep "^world" | sed "s/^/$v: /"
done | tee "$OUT/ab_head_strip8.txt"

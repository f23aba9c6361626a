#!/bin/bash
# Round 5: ownership reads the boxes' standing values a phase early - tests, phase timers, A/B on the launch-bound cases
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zb}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "owner or region or strip or native or trusted or kept" > "$OUT/pytest_own.log" 2>&1
tail -2 "$OUT/pytest_own.log"
grep -q " passed" "$OUT/pytest_own.log" || { tail -60 "$OUT/pytest_own.log"; exit 1; }
grep -q "failed" "$OUT/pytest_own.log" && { tail -80 "$OUT/pytest_own.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
for c in cfg3 cfg2; do
  PANO_LIB=$PWD/build/variants/ow_stamp/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep "sampled\|cycles"
  for r in 1 2; do
  timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep -v amdgpu.ids
  PANO_LIB=$PWD/build/variants/own_prev/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep -v amdgpu.ids | sed "s/^/before: /"
  done
done | tee "$OUT/own_box_prefetch.txt"
grep -q "GPU core dump" "$OUT/own_box_prefetch.txt" && exit 1
for r in 1 2; do for v in own_prev base; do
  if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
  echo "$v: $(PANO_LIB=$LIB PANO_STRIP_RANK=4 PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=1 timeout -k 10 300 python tools/strip_floor.py cfg3 8 2>/dev/null | grep '^world 8: slowest' | cut -c1-330)"
done; done | tee "$OUT/own_box_prefetch_strip.txt"
exit 0

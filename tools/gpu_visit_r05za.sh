#!/bin/bash
# Round 5: ownership's write-out as 16-byte pieces, boxes and marks per quarter - tests, phase timers, A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05za}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "owner or region or strip or stitch or native or trusted or kept or properties or cfg2 or cfg5 or valid" > "$OUT/pytest_own.log" 2>&1
tail -2 "$OUT/pytest_own.log"
grep -q " passed" "$OUT/pytest_own.log" || { tail -60 "$OUT/pytest_own.log"; exit 1; }
grep -q "failed" "$OUT/pytest_own.log" && { tail -80 "$OUT/pytest_own.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
timeout -k 10 600 python tools/fuzz_ownership.py 40 > "$OUT/fuzz.txt" 2>&1; tail -2 "$OUT/fuzz.txt"
grep -q "GPU core dump" "$OUT/fuzz.txt" && exit 1
for c in cfg3 cfg5 cfg2; do
  PANO_LIB=$PWD/build/variants/ow_stamp/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep "sampled\|cycles"
  timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep -v amdgpu.ids
  PANO_LIB=$PWD/build/variants/own_prev/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep -v amdgpu.ids | sed "s/^/before: /"
done | tee "$OUT/own_writeout.txt"
grep -q "GPU core dump" "$OUT/own_writeout.txt" && exit 1
tools/ab_libs.sh cfg3 3 own_prev base | tee "$OUT/ab_own_writeout_cfg3.txt"
tools/ab_libs.sh cfg2 2 own_prev base | tee "$OUT/ab_own_writeout_cfg2.txt"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 300 python tools/strip_floor.py cfg3 1 8 2>/dev/null | grep "^world" | cut -c1-200
exit 0

#!/bin/bash
# Round 5: ownership - a sub-tile with one certain owner skips the second level of bounds
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05t}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
for c in cfg3 cfg5 cfg2; do PANO_LIB=$PWD/build/variants/ow_stamp/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep "sampled\|cycles";
  timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep -v amdgpu.ids
  PANO_OWN_PRUNE=3 timeout -k 10 200 python tools/probe_own_stamps.py $c 2>/dev/null | sed "s/^/round 4 kernel: /"
done | tee "$OUT/own_subfill.txt"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 300 python tools/strip_floor.py cfg3 1 8 2>/dev/null | grep "^world"
exit 0

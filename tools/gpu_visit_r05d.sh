#!/bin/bash
# Round 5, fourth visit: the ownership kernel with dynamic LDS, staged cameras and OW_ILP pixels per
# lane - parity subset, phase timers, ILP 1 / 2 / 4.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05d}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu (ownership, regions, strips, full size)"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "ownership or regions or strip or full_size or cameras or rccl or golden or stitch" > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || exit 1
grep -q "failed" "$OUT/pytest_gpu.log" && exit 1
echo "== ownership: ILP variants, phase timers, no evaluation"
for c in cfg3 cfg5 cfg2; do
for v in base ow_ilp1 ow_ilp4 ow_noeval ow_stamp; do
  if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
  PANO_LIB=$LIB timeout -k 10 200 python tools/probe_own_stamps.py $c 2>/dev/null | sed "s/^/$v: /" | tee -a "$OUT/own_stamps.txt"
done
PANO_OWN_PRUNE=3 timeout -k 10 200 python tools/probe_own_stamps.py $c 2>/dev/null | sed "s/^/round 4 kernel: /" | tee -a "$OUT/own_stamps.txt"
done

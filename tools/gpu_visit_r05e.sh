#!/bin/bash
# Round 5, fifth visit: ownership tile height (OW_SUBS 4 / 8 / 16), ILP 4, no evaluation; what the
# warp waits for (ablations: no table look-ups / no stores / no frame reads; 2 / 8 rows per thread).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05e}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
fault() { grep -l "GPU core dump\|Memory access fault" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null && { echo "GPU FAULT"; exit 1; }; }
echo "== pytest -m gpu (ownership, regions, strips, full size)"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "ownership or regions or strip or full_size or cameras or rccl or golden or stitch" > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || exit 1
grep -q "failed" "$OUT/pytest_gpu.log" && exit 1
fault
echo "== ownership: tile height, ILP, no evaluation"
for c in cfg3 cfg5 cfg2; do
for v in base ow_subs4 ow_subs16 ow_subs8_ilp4 ow_noeval; do
  if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
  PANO_LIB=$LIB timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /" | tee -a "$OUT/own_variants.txt"
  fault
done
PANO_OWN_PRUNE=3 timeout -k 10 200 python tools/probe_own_stamps.py $c 2>/dev/null | sed "s/^/round 4 kernel: /" | tee -a "$OUT/own_variants.txt"
done
echo "== warp ablations (timing only)"
tools/ab_libs.sh cfg3 2 base warp_nolut warp_nostore warp_noload warp_rows2 warp_rows8 | tee "$OUT/ab_warp_ablations_cfg3.txt"
fault

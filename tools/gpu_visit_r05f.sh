#!/bin/bash
# Round 5, sixth visit: full GPU tests of the tree (ownership 64 x 128 tiles, ILP 4); the region
# search fused into the ownership kernel on small mosaics (config 2, a world-8 strip); the
# collapse with every plane read line-aligned (timing only); warp with non-temporal stores; the
# ownership kernel's phase timers (private rows, no atomics).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05f}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
fault() { if grep -l "GPU core dump\|Memory access fault" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null; then echo "GPU FAULT"; exit 1; fi; return 0; }
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || exit 1
grep -q "failed" "$OUT/pytest_gpu.log" && exit 1
fault
echo "== ownership phase timers"
for c in cfg3 cfg5; do
  PANO_LIB=$PWD/build/variants/ow_stamp/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/own_stamps.txt"
done
fault
echo "== region search fused into the ownership kernel (PANO_REGIONS_FUSED) on config 2"
tools/ab_env.sh cfg2 3 PANO_REGIONS_FUSED 0 1 | tee "$OUT/ab_regions_fused_cfg2.txt"
echo "== ... and on a world-8 / world-4 strip of config 3 (two lanes, plan from the memo)"
for v in 0 1; do
  PANO_REGIONS_FUSED=$v PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=2 timeout -k 10 300 python tools/strip_floor.py cfg3 4 8 2>/dev/null | grep "^world" | sed "s/^/fused=$v: /"
done | tee "$OUT/ab_regions_fused_strips.txt"
fault
echo "== collapse with line-aligned plane reads (timing only) / warp with nt stores"
tools/ab_libs.sh cfg3 3 base compose_aligned warp_nt | tee "$OUT/ab_compose_aligned_warp_nt_cfg3.txt"
tools/ab_libs.sh cfg5 1 base compose_aligned | tee "$OUT/ab_compose_aligned_cfg5.txt"
fault

#!/bin/bash
# Round 5, tenth visit: full GPU tests; ownership tile height by grid size (PANO_OWN_SMALL_BELOW:
# 0 = always 64 x 128, 1536 = 64 x 64 below 1536 workgroups) on strips and config 2.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05j}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
fault() { if grep -l "GPU core dump\|Memory access fault" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null; then echo "GPU FAULT"; exit 1; fi; return 0; }
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
fault
echo "== and with the small tiles forced on every size (PANO_OWN_SMALL_BELOW huge): ownership tests"
PANO_OWN_SMALL_BELOW=100000000 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "ownership or regions or cameras or cfg5_full_size or golden" > "$OUT/pytest_small_tiles.log" 2>&1; tail -2 "$OUT/pytest_small_tiles.log"
grep -q "failed" "$OUT/pytest_small_tiles.log" && { tail -60 "$OUT/pytest_small_tiles.log"; exit 1; }
fault
echo "== strips, three lanes trusted: tile height by grid"
for b in 0 1536 0 1536; do
  PANO_OWN_SMALL_BELOW=$b PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 300 python tools/strip_floor.py cfg3 4 8 2>/dev/null | grep "^world" | sed "s/^/small below $b: /" | sed "s/{.*ownership_cameras_kernel/... ownership_cameras_kernel/"
  fault
done | tee "$OUT/own_small_strips.txt"
echo "== config 2"
tools/ab_env.sh cfg2 3 PANO_OWN_SMALL_BELOW 0 1536 | tee "$OUT/own_small_cfg2.txt"
for b in 0 1536; do PANO_OWN_SMALL_BELOW=$b timeout -k 10 200 python tools/probe_own_stamps.py cfg2 2>/dev/null | sed "s/^/small below $b: /"; done | tee -a "$OUT/own_small_cfg2.txt"
fault

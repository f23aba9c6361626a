#!/bin/bash
# Round 5: the interior-map kernels on a strip's blocks only - full tests, strip floors, config 2.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05n}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
echo "== strips: one lane (the chain) and three lanes"
for l in 1 3; do
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=$l timeout -k 10 300 python tools/strip_floor.py cfg3 1 4 8 2>/dev/null | grep "^world" | sed "s/^/lanes $l: /"
done | tee "$OUT/strips.txt"
tools/gpu_profile.sh "$T" cfg2 30 | grep -v "^at::\|rocclr"
grep -l "GPU core dump" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null && exit 1
exit 0

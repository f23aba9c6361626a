#!/bin/bash
# Round 5: the collapse takes an interior pixel's colour out of its owner's warped planes (part 3) instead of
# sampling the frame again - all GPU tests, then A/B against the sampling form (compose_shade)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zd}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -80 "$OUT/pytest_gpu.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
tools/ab_libs.sh cfg3 3 compose_shade base | tee "$OUT/ab_interior_from_planes_cfg3.txt"
tools/ab_libs.sh cfg5 1 compose_shade base | tee "$OUT/ab_interior_from_planes_cfg5.txt"
tools/ab_libs.sh cfg2 2 compose_shade base | tee "$OUT/ab_interior_from_planes_cfg2.txt"
grep -l "GPU core dump" gpurun_out/ab/*.txt 2>/dev/null && exit 1
exit 0

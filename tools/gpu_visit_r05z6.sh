#!/bin/bash
# Round 5: strips of equal work against strips of equal width (config 3, every rank of world 2 / 4 / 8)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05z6}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column_strips or sharded or kept or trusted" > "$OUT/pytest_strips.log" 2>&1
tail -2 "$OUT/pytest_strips.log"
grep -q " passed" "$OUT/pytest_strips.log" || { tail -60 "$OUT/pytest_strips.log"; exit 1; }
grep -q "failed" "$OUT/pytest_strips.log" && { tail -80 "$OUT/pytest_strips.log"; exit 1; }
for keep in 0 1; do for bal in 0 1; do
  PANO_STRIP_BALANCE=$bal PANO_KEEP_GEOMETRY=$keep PANO_SETS_IN_FLIGHT=3 PANO_PLAN_CACHED=1 timeout -k 10 600 python tools/strip_floor.py cfg3 1 2 4 8 > "$OUT/strip_keep${keep}_bal$bal.txt" 2>&1 || { tail -30 "$OUT/strip_keep${keep}_bal$bal.txt"; exit 1; }
  grep -q "GPU core dump" "$OUT/strip_keep${keep}_bal$bal.txt" && exit 1
  echo "== kept geometry $keep, balanced $bal"; grep -E "^world" "$OUT/strip_keep${keep}_bal$bal.txt" | cut -c1-170
done; done | tee "$OUT/balanced_strips_cfg3.txt"
exit 0

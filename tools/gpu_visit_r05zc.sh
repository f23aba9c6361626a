#!/bin/bash
# Round 5: the kept-geometry tests (equalised colour tables, lanes of a ShardedStitcher)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zc}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > "$OUT/pytest_parity.log" 2>&1
tail -3 "$OUT/pytest_parity.log"
grep -q " passed" "$OUT/pytest_parity.log" || { tail -60 "$OUT/pytest_parity.log"; exit 1; }
grep -q "failed" "$OUT/pytest_parity.log" && { tail -80 "$OUT/pytest_parity.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
exit 0

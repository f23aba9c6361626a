#!/bin/bash
# Round 5: the whole GPU suite and the smoke on the round's last tree
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zr}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -80 "$OUT/pytest_gpu.log"; exit 1; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
exit 0

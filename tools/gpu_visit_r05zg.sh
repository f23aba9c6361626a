#!/bin/bash
# Round 5: config 2 with one to four stitches in flight
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zg}; mkdir -p "$OUT"; export TMPDIR=/tmp
for r in 1 2; do for n in 1 2 3 4; do
  timeout -k 10 300 python bench.py --workload cfg2 --steps 60 --warmup 6 --in-flight $n --no-cpu-baseline --no-secondary --busy-seconds 0 2>"$OUT/err.txt" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 in flight $n: %.4f ms per stitch' % d['ms_per_step'])"
done; done | tee "$OUT/cfg2_in_flight.txt"
grep -q "GPU core dump" "$OUT/err.txt" && exit 1
exit 0

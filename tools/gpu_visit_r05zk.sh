#!/bin/bash
# Round 5: clocks and power while the config-3 stitch runs (the blur's time moves by 15 % from run to run)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zk}; mkdir -p "$OUT"; export TMPDIR=/tmp
for r in 1 2 3; do
timeout -k 10 300 python tools/probe_clocks.py -- python bench.py --workload cfg3 --steps 600 --warmup 5 --no-cpu-baseline --no-secondary --busy-seconds 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.rstrip()
    if l.startswith('{'):
        d = json.loads(l[:l.rfind('}') + 1]) if l.endswith('}') else None
        print('bench line (cut)', l[:120])
    else:
        print(l)
"
done | tee "$OUT/clocks_cfg3.txt"
exit 0

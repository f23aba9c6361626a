#!/bin/bash
# Round 5: the warp on the need flags (PANO_WARP_NEED=1: only the blocks of a window anything reads)
# on config 3 / 5, now that the side chain is cheap
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05q}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
for wl in cfg3 cfg5; do
for r in 1 2; do for v in 0 1; do
PANO_WARP_NEED=$v timeout -k 10 300 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --busy-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$wl warp_need=$v step %.4f one-in-flight %s warp %.4f blur %.4f tile_flags %.4f' % (d['ms_per_step'], d.get('ms_per_stitch_one_in_flight'), k.get('warp_windows_kernel', 0), k.get('blur_lean_kernel', 0) + k.get('blur_lean5_kernel', 0), k.get('tile_flags_kernel', 0)))"
done; done; done | tee "$OUT/ab_warp_need.txt"
exit 0

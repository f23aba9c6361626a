#!/bin/bash
# Round 5: the side stream inside a stitch on small mosaics too (PANO_TWO_STREAMS_MIN_PX = 0)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05p}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== strips (plan memo, trusted): one stream / two streams inside a stitch, 1 and 3 lanes"
for px in 16777216 0 16777216 0; do for l in 1 3; do
PANO_TWO_STREAMS_MIN_PX=$px PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=$l timeout -k 10 300 python tools/strip_floor.py cfg3 8 2>/dev/null | grep "^world" | sed "s/^/min_px $px lanes $l: /" | sed "s/(timed.*//"
done; done | tee "$OUT/two_streams_strips.txt"
echo "== config 2"
tools/ab_env.sh cfg2 3 PANO_TWO_STREAMS_MIN_PX 16777216 0 | tee "$OUT/two_streams_cfg2.txt"
grep -l "GPU core dump" "$OUT"/*.txt 2>/dev/null && exit 1
exit 0

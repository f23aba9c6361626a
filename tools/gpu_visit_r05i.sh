#!/bin/bash
# Round 5, ninth visit: the blur's segment length chosen by a list-schedule estimate - strips'
# parity, the model's choice against forced lengths on world-8 / world-4 strips (ranks 0, N/2,
# N-1 are sampled by the tool; the slowest is printed), configs 2 / 3 / 5 against the round's
# first commit; the blur's A/B builds through the parity tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05i}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
fault() { if grep -l "GPU core dump\|Memory access fault" "$OUT"/*.txt "$OUT"/*.log 2>/dev/null; then echo "GPU FAULT"; exit 1; fi; return 0; }
echo "== pytest -m gpu (strips, segments, blur)"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "strip or segment or blur or full_size or closed_360 or trusted or rccl" > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
grep -q " passed" "$OUT/pytest_gpu.log" || { tail -40 "$OUT/pytest_gpu.log"; exit 1; }
grep -q "failed" "$OUT/pytest_gpu.log" && { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
fault
for w in 8 4; do
echo "== world-$w strip: the model (T=0) against forced segment lengths, one lane"
for t in 0 -1 16 20 24 32 40 48 64; do
  PANO_BLUR_SEG_T=$t PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=1 timeout -k 10 300 python tools/strip_floor.py cfg3 $w 2>/dev/null | grep "^world" | sed "s/^/T=$t: /" | sed "s/{.*blur_lean_kernel/... blur_lean_kernel/"
  fault
done | tee "$OUT/blur_seg_t_strip$w.txt"
done
echo "== three lanes, trusted, world 8 and world 1 2 4 8 with two"
PANO_PLAN_CACHED=1 PANO_SETS_IN_FLIGHT=3 timeout -k 10 300 python tools/strip_floor.py cfg3 1 2 4 8 --json=$OUT/strip_floor_cfg3_three_lanes_trusted.json 2>/dev/null | grep "^world" | tee "$OUT/strip_floor_3lanes.txt"
fault
echo "== one GPU against the round's first commit"
tools/ab_libs.sh cfg3 2 r05head base | tee "$OUT/ab_head_cfg3.txt"
tools/ab_libs.sh cfg2 2 r05head base | tee "$OUT/ab_head_cfg2.txt"
tools/ab_libs.sh cfg5 1 r05head base | tee "$OUT/ab_head_cfg5.txt"
fault
echo "== the blur's A/B builds through the parity tests"
bash tools/gpu_variants_parity.sh $T
fault

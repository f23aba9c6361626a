#!/bin/bash
# Round 5: the scale step on 64 x 64 tiles where a plane fills the chip with them - tests, A/B on config 4
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zi}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "scale or pyr or sift or cfg4 or gaussian or keypoint or octave or dog" > "$OUT/pytest_scale.log" 2>&1
tail -3 "$OUT/pytest_scale.log"
grep -q " passed" "$OUT/pytest_scale.log" || { tail -60 "$OUT/pytest_scale.log"; exit 1; }
grep -q "failed" "$OUT/pytest_scale.log" && { tail -80 "$OUT/pytest_scale.log"; exit 1; }
grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
for r in 1 2 3; do tools/ab_cfg4.sh ss_tall0 base; done | tee "$OUT/ab_scale_tall_cfg4.txt"
exit 0

#!/bin/bash
# The blur's A/B builds still agree with the default one: -DMB_STREAM_EDGE=0 (regular items through
# ms_body, the ones with reflected columns / unaligned windows through blur_irregular_kernel beside
# it) and -DMB_STREAM=0 (round 3's ml_body), each through the blur's parity tests - planes against
# the oracle and the reference-run golden, lean against general, windows, strips.  The variants are
# built here (tools/build_variant.sh) and travel with the snapshot.   tools/gpu_variants_parity.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-variants}
mkdir -p "$OUT"
for v in blur_edge0 blur_stream0; do
  echo "== $v"
  PANO_LIB=$PWD/build/variants/$v/libpano360_hip.so timeout -k 10 600 python -m pytest tests -m gpu -x -q \
      -k "lean_blur or blur_planes or windowed_blur or fused_windows or closed_360 or column_strips or interior_shortcut or cfg2_full_size" \
      > "$OUT/pytest_$v.log" 2>&1
  tail -3 "$OUT/pytest_$v.log"
  grep -q "GPU core dump" "$OUT/pytest_$v.log" && exit 1
done
exit 0

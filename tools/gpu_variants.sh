#!/bin/bash
# Correctness (blur-related parity tests) then alternating A/B timing of library variants.
#   tools/gpu_variants.sh TAG WORKLOAD REPS name1 name2 ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; WL=$2; REPS=$3; shift; shift; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for v in "$@"; do
  [ "$v" = base ] && continue
  PANO_LIB=$PWD/build/variants/$v/libpano360_hip.so timeout -k 10 600 python -m pytest tests -m gpu -x -q \
      -k "blur_planes or windowed_blur or fused_windows or level_counts or interior_shortcut or column_strips or closed_360" \
      > "$OUT/pytest_$v.log" 2>&1
  echo "$v: $(tail -1 "$OUT/pytest_$v.log")"
done
bash tools/ab_libs.sh "$WL" "$REPS" "$@" | tee "$OUT/ab_$WL.txt"

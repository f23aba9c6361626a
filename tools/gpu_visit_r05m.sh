#!/bin/bash
# where the sort kernel's estimate spends its time: builds without the histograms / the order statistics
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05m}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in base; do
  if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
  echo "-- $v"
  PANO_LIB=$LIB tools/gpu_profile.sh "$T" cfg2 20 | grep "mb_sort"
done
exit 0

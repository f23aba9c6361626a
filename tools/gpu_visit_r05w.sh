#!/bin/bash
# Round 5: the collapse's interior pixels as an LDS-free kernel on the side stream beside the blur
# (split_global: shared table read from global memory; split_identity: v / 255 computed) - parity of
# both builds on the stitch tests, then A/B against the one-kernel collapse
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r05w}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in base split_global split_identity; do
  if [ "$v" = base ]; then LIB=""; else LIB=$PWD/build/variants/$v/libpano360_hip.so; fi
  PANO_LIB=$LIB timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "stitch or strip or native or trusted or collapse or compose or window or properties" > "$OUT/pytest_$v.log" 2>&1
  echo "$v: $(tail -1 "$OUT/pytest_$v.log")"
  grep -q " passed" "$OUT/pytest_$v.log" || { tail -40 "$OUT/pytest_$v.log"; exit 1; }
  grep -q "failed" "$OUT/pytest_$v.log" && { tail -60 "$OUT/pytest_$v.log"; exit 1; }
  grep -l "GPU core dump" "$OUT"/*.log 2>/dev/null && exit 1
done
tools/ab_libs.sh cfg3 3 base split_global split_identity | tee "$OUT/ab_split_cfg3.txt"
tools/ab_libs.sh cfg5 1 base split_global split_identity | tee "$OUT/ab_split_cfg5.txt"
tools/ab_libs.sh cfg2 2 base split_global split_identity | tee "$OUT/ab_split_cfg2.txt"
grep -l "GPU core dump" gpurun_out/ab/*.txt 2>/dev/null && exit 1
exit 0

#!/bin/bash
# Round 5: the ownership kernel's phase timers on the round's last code
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05z9}; mkdir -p "$OUT"; export TMPDIR=/tmp
for c in cfg3 cfg5 cfg2; do
  PANO_LIB=$PWD/build/variants/ow_stamp/libpano360_hip.so timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep "sampled\|cycles"
  timeout -k 10 200 python tools/probe_own_stamps.py $c 2>&1 | grep -v amdgpu.ids
done | tee "$OUT/own_stamps_final.txt"
grep -q "GPU core dump" "$OUT/own_stamps_final.txt" && exit 1
exit 0

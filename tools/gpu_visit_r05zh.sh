#!/bin/bash
# Round 5: the host side of a config-2 stitch under cProfile
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zh}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 python tools/host_profile.py cfg2 > "$OUT/host_profile_cfg2.txt" 2>&1 || { tail -20 "$OUT/host_profile_cfg2.txt"; exit 1; }
grep -q "GPU core dump" "$OUT/host_profile_cfg2.txt" && exit 1
head -60 "$OUT/host_profile_cfg2.txt"

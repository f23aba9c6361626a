#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05z5}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 python tools/probe_strip_pixels.py cfg3 1 2 4 8 > "$OUT/strip_pixels_cfg3.txt" 2>&1 || { tail -20 "$OUT/strip_pixels_cfg3.txt"; exit 1; }
grep -q "GPU core dump" "$OUT/strip_pixels_cfg3.txt" && exit 1
cat "$OUT/strip_pixels_cfg3.txt"

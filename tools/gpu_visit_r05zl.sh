#!/bin/bash
# Round 5: the blur's time against where its arenas start (offsets into the allocation; other allocations in front)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05zl}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python ${PROBE:-tools/probe_arena_skew.py} cfg3 > "$OUT/arena_skew_cfg3.txt" 2>&1 || { tail -20 "$OUT/arena_skew_cfg3.txt"; exit 1; }
grep -q "GPU core dump" "$OUT/arena_skew_cfg3.txt" && exit 1
grep -v amdgpu.ids "$OUT/arena_skew_cfg3.txt" | cut -c1-250

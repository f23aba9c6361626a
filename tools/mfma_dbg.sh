#!/bin/bash
# time the matrix-core blur with parts switched off (1 stores, 2 fetch, 4 commit, 8 mfma,
# 16 stores into a 64 KiB region, 32 loads from a small region)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for d in ${@:-0 1 2 4 8 3 7 15 14}; do
  PANO_MFMA_DBG=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg $d', round(d['kernel_ms_per_step']['blur_mfma_kernel'],3))"
done

#!/bin/bash
# blur_mfma_kernel with parts switched off (PANO_MFMA_DBG bits: 1 stores, 2 fetch, 4 commit,
# 8 MFMAs): timing experiments only, the mosaics are wrong
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ablate
for d in ${@:-0 1 2 4 8 3 5 6 7 9 10 12 15}; do
  PANO_MFMA_DBG=$d timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg=$d blur %.3f ms  step %.3f' % (d['kernel_ms_per_step']['blur_mfma_kernel'], d['ms_per_step']))"
done | tee gpurun_out/ablate/ablate.txt

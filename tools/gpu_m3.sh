#!/bin/bash
# the twelve-wave form of the blur: parity tests that exercise the multiband path, then A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/m3
PANO_BLUR=mfma3 timeout 900 python -m pytest tests -m gpu -x -q -k "multiband or fused or strips or interior or sweep or degenerate or level_counts or full_size or stitch_entry or warp_need" > gpurun_out/m3/pytest.log 2>&1; tail -5 gpurun_out/m3/pytest.log
bash tools/gpu_ab2.sh cfg3 PANO_BLUR mfma mfma3 mfma mfma3

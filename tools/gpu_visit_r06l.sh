#!/bin/bash
# Round 6: a row's two taps by ONE 8-byte load (was 4 + 2 bytes) in the warp, the collapse's interior
# pixels and the fused linear / none blends - parity (every sampling test), then A/B against -DTAPS_LOAD64=0.
#   tools/gpu_visit_r06l.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06l}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest (everything that samples frames)"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not sift and not knn and not crop and not laplacian and not cfg4" > "$OUT/pytest.log" 2>&1; rc=$?; tail -3 "$OUT/pytest.log"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest.log" | head -30; exit 1; }
for wl in cfg3 cfg2 cfg5; do
  steps=30; [ $wl = cfg5 ] && steps=6
  for rep in 1 2 3 4; do
    for v in base taps32; do
      L=""; [ $v = base ] || L=$PWD/build/variants/$v/libpano360_hip.so
      PANO_LIB=$L timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --busy-seconds 0 --side-file "$OUT/ab_${wl}_${v}_$rep.json" > /dev/null 2> "$OUT/ab.err" || { tail -5 "$OUT/ab.err"; exit 1; }
    done
  done
  python - "$OUT" $wl <<'P'
import json, sys, statistics as st
out, wl = sys.argv[1:3]
for v in ("base", "taps32"):
    ms, comp, warp = [], [], []
    for rep in (1, 2, 3, 4):
        d = json.load(open(f"{out}/ab_{wl}_{v}_{rep}.json")); k = d["kernel_ms_per_step"]
        ms.append(d["ms_per_step"]); comp.append(k.get("multiband_compose_kernel", 0)); warp.append(k.get("warp_windows_kernel", 0))
    print(f"{wl} {v:7s}: ms/step median {st.median(ms):.3f} {['%.3f' % x for x in ms]}  warp {st.median(warp):.4f} {['%.4f' % x for x in warp]}  collapse {st.median(comp):.4f}")
P
done 2>&1 | tee "$OUT/ab_taps_load64.txt"

#!/bin/bash
# Round 6: the collapse's gathers as buffer loads off a scalar record (was: flat loads at per-lane
# addresses) - parity, then the working tree's library against the commit before it, alternating.
#   tools/gpu_visit_r06j.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06j}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest (everything that stitches)"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not sift and not knn and not crop and not laplacian and not cfg4" > "$OUT/pytest.log" 2>&1; rc=$?; tail -3 "$OUT/pytest.log"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest.log" | head -30; exit 1; }
for wl in cfg3 cfg2 cfg5; do
  steps=30; [ $wl = cfg5 ] && steps=6
  for rep in 1 2 3 4; do
    for v in base flatgather; do
      L=""; [ $v = base ] || L=$PWD/build/variants/$v/libpano360_hip.so
      PANO_LIB=$L timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --busy-seconds 0 --side-file "$OUT/ab_${wl}_${v}_$rep.json" > /dev/null 2> "$OUT/ab.err" || { tail -5 "$OUT/ab.err"; exit 1; }
    done
  done
  python - "$OUT" $wl <<'P'
import json, sys, statistics as st
out, wl = sys.argv[1:3]
for v in ("base", "flatgather"):
    ms, comp, one = [], [], []
    for rep in (1, 2, 3, 4):
        d = json.load(open(f"{out}/ab_{wl}_{v}_{rep}.json")); k = d["kernel_ms_per_step"]
        ms.append(d["ms_per_step"]); comp.append(k.get("multiband_compose_kernel", 0)); one.append(d.get("ms_per_stitch_one_in_flight") or 0)
    print(f"{wl} {v:10s}: ms/step median {st.median(ms):.3f} {['%.3f' % x for x in ms]}  collapse {st.median(comp):.4f} {['%.4f' % x for x in comp]}  one in flight {st.median(one):.3f}")
P
done 2>&1 | tee "$OUT/ab_collapse_buffer_gathers.txt"

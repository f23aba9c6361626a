#!/bin/bash
# Round 6: the collapse's tiles sorted by kind (PANO_COMPOSE_COMPACT) - parity, then A/B on one box.
#   tools/gpu_visit_r06g.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06g}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest (the option's own test; then the stitch tests with the option forced on)"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "sorted_by_kind or trusted_stitches" > "$OUT/pytest_compact.log" 2>&1; rc=$?; tail -3 "$OUT/pytest_compact.log"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest_compact.log" | head -30; exit 1; }
PANO_COMPOSE_COMPACT=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not sift and not knn and not crop and not laplacian and not cfg4" > "$OUT/pytest_compact_forced.log" 2>&1; rc=$?; tail -3 "$OUT/pytest_compact_forced.log"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest_compact_forced.log" | head -30; exit 1; }
for wl in cfg3 cfg2 cfg5; do
  steps=30; [ $wl = cfg5 ] && steps=6
  for rep in 1 2 3; do
    for on in 1 0; do
      PANO_COMPOSE_COMPACT=$on timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --busy-seconds 0 --side-file "$OUT/ab_${wl}_${on}_$rep.json" > /dev/null 2> "$OUT/ab.err" || { tail -5 "$OUT/ab.err"; exit 1; }
    done
  done
  python - "$OUT" $wl <<'P'
import json, sys, statistics as st
out, wl = sys.argv[1:3]
for on in ("1", "0"):
    ms, comp, one = [], [], []
    for rep in (1, 2, 3):
        d = json.load(open(f"{out}/ab_{wl}_{on}_{rep}.json"))
        k = d["kernel_ms_per_step"]
        ms.append(d["ms_per_step"]); comp.append(k.get("multiband_compose_kernel", 0)); one.append(d.get("ms_per_stitch_one_in_flight") or 0)
    print(f"{wl} sorted tiles {'on ' if on == '1' else 'off'}: ms/step median {st.median(ms):.3f} {['%.3f' % v for v in ms]}  collapse {st.median(comp):.4f} {['%.4f' % v for v in comp]}  one in flight {st.median(one):.3f}")
P
done 2>&1 | tee "$OUT/ab_compose_compact.txt"

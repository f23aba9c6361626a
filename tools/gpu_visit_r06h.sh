#!/bin/bash
# Round 6: the collapse again, on one box - level classes and sorted tiles as template / option, four
# variants alternating.   tools/gpu_visit_r06h.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06h}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest (classes, sorted tiles, shortcut, strips)"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "sorted_by_kind or level_classes or interior_shortcut or column_strips or fused" > "$OUT/pytest.log" 2>&1; rc=$?; tail -3 "$OUT/pytest.log"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest.log" | head -30; exit 1; }
for wl in cfg3 cfg2 cfg5; do
  steps=30; [ $wl = cfg5 ] && steps=6
  for rep in 1 2 3; do
    for v in 00 10 01 11; do
      PANO_COMPOSE_COMPACT=${v:0:1} PANO_LEVEL_CLASSES=${v:1:1} timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --busy-seconds 0 --side-file "$OUT/ab_${wl}_${v}_$rep.json" > /dev/null 2> "$OUT/ab.err" || { tail -5 "$OUT/ab.err"; exit 1; }
    done
  done
  python - "$OUT" $wl <<'P'
import json, sys, statistics as st
out, wl = sys.argv[1:3]
for v in ("00", "10", "01", "11"):
    ms, comp = [], []
    for rep in (1, 2, 3):
        d = json.load(open(f"{out}/ab_{wl}_{v}_{rep}.json"))
        ms.append(d["ms_per_step"]); comp.append(d["kernel_ms_per_step"].get("multiband_compose_kernel", 0))
    print(f"{wl} sorted tiles {v[0]} level classes {v[1]}: ms/step median {st.median(ms):.3f} {['%.3f' % x for x in ms]}  collapse {st.median(comp):.4f} {['%.4f' % x for x in comp]}")
P
done 2>&1 | tee "$OUT/ab_collapse_variants.txt"

#!/bin/bash
# Round 6: level classes in the collapse - parity tests, then A/B of config 3 / 2 / 5 with the classes
# on and off (alternating, same box), then the counter passes for the collapse's traffic.
#   tools/gpu_visit_r06e.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06e}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest (classes, shortcut, strips, kept geometry, fused paths)"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "level_classes or interior_shortcut or column_strips or kept_geometry or fused or closed_360 or trusted or stitch_entry or cfg2_full" > "$OUT/pytest_classes.log" 2>&1; rc=$?; tail -4 "$OUT/pytest_classes.log"
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" "$OUT/pytest_classes.log" | head -30; exit 1; }
for wl in cfg3 cfg2 cfg5; do
  steps=30; [ $wl = cfg5 ] && steps=6
  for rep in 1 2 3; do
    for on in 1 0; do
      PANO_LEVEL_CLASSES=$on timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --busy-seconds 0 --side-file "$OUT/ab_${wl}_${on}_$rep.json" > /dev/null 2> "$OUT/ab.err" || { tail -5 "$OUT/ab.err"; exit 1; }
    done
  done
  python - "$OUT" $wl <<'P'
import json, sys, statistics as st
out, wl = sys.argv[1:3]
for on in ("1", "0"):
    ms, comp, blur, gb = [], [], [], []
    for rep in (1, 2, 3):
        d = json.load(open(f"{out}/ab_{wl}_{on}_{rep}.json"))
        k = d["kernel_ms_per_step"]
        ms.append(d["ms_per_step"]); comp.append(k.get("multiband_compose_kernel", 0))
        blur.append(sum(v for n, v in k.items() if n.startswith("blur_")))
        gb.append(d.get("collapse_gather_GB"))
    print(f"{wl} classes {'on ' if on == '1' else 'off'}: ms/step median {st.median(ms):.3f} {['%.3f' % v for v in ms]}  collapse {st.median(comp):.3f} {['%.3f' % v for v in comp]}  blur {st.median(blur):.3f}  gather {gb[0]:.3f} GB")
P
done 2>&1 | tee "$OUT/ab_level_classes.txt"
echo "== counters of config 3 (the collapse's traffic)"
tools/pmc.sh "$T/pmc_cfg3" cfg3 > "$OUT/pmc_cfg3.log" 2>&1; tail -1 "$OUT/pmc_cfg3.log"
grep -A12 "== multiband_compose_kernel" "$OUT/pmc_cfg3/summary.txt" | head -16

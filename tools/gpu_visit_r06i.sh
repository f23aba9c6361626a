#!/bin/bash
# Round 6: is the default collapse slower than round 5's?  Same box, alternating: the library as it is
# (sorted tiles off / on) against a build without the compaction code (round 5's prologue).
#   tools/gpu_visit_r06i.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06i}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
for wl in cfg3 cfg5; do
  steps=30; [ $wl = cfg5 ] && steps=6
  for rep in 1 2 3 4; do
    for v in base0 base1 nocompact; do
      case $v in
        base0) L=""; C=0;; base1) L=""; C=1;; nocompact) L=$PWD/build/variants/nocompact/libpano360_hip.so; C=0;;
      esac
      PANO_LIB=$L PANO_COMPOSE_COMPACT=$C timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --busy-seconds 0 --side-file "$OUT/ab_${wl}_${v}_$rep.json" > /dev/null 2> "$OUT/ab.err" || { tail -5 "$OUT/ab.err"; exit 1; }
    done
  done
  python - "$OUT" $wl <<'P'
import json, sys, statistics as st
out, wl = sys.argv[1:3]
for v in ("base0", "base1", "nocompact"):
    ms, comp, warp, blur = [], [], [], []
    for rep in (1, 2, 3, 4):
        d = json.load(open(f"{out}/ab_{wl}_{v}_{rep}.json")); k = d["kernel_ms_per_step"]
        ms.append(d["ms_per_step"]); comp.append(k.get("multiband_compose_kernel", 0)); warp.append(k.get("warp_windows_kernel", 0))
        blur.append(sum(x for n, x in k.items() if n.startswith("blur_")))
    print(f"{wl} {v:9s}: ms/step median {st.median(ms):.3f}  collapse {st.median(comp):.4f} {['%.4f' % x for x in comp]}  warp {st.median(warp):.4f}  blur {st.median(blur):.4f}")
P
done 2>&1 | tee "$OUT/ab_collapse_prologue.txt"

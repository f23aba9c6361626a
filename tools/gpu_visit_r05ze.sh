#!/bin/bash
# Round 5: 4-rank dry run of the strips bench on the one GPU (gloo, host-staged exchange): balanced bounds
# broadcast from rank 0, three lanes per rank, the secondaries (reduce exchange, per-stitch plan, kept geometry)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r05ze}; mkdir -p "$OUT"; export TMPDIR=/tmp
PANO_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 4 --steps 3 --warmup 1 > "$OUT/bench_4rank_selflaunch.json" 2> "$OUT/bench_4rank.err" || { tail -30 "$OUT/bench_4rank.err"; exit 1; }
python - "$OUT/bench_4rank_selflaunch.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["n_gpus"], d["ms_per_step"], d["config"]["parallelism"][-120:])
for k, v in d.get("secondary", {}).items():
    print(" ", k, {kk: v[kk] for kk in v if kk != "what"} if isinstance(v, dict) else v)
P
grep -l "GPU core dump" "$OUT"/*.err 2>/dev/null && exit 1
exit 0

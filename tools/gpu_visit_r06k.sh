#!/bin/bash
# Round 6: dry runs of bench.py's multi-rank paths on one GPU (gloo): the other --plan, the other
# exchange, config 5 and config 2 as strips, config 4 as replicas, three ranks.   tools/gpu_visit_r06k.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r06k}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp PANO_DIST_BACKEND=gloo
run() {  # tag, args...
  tag=$1; shift
  timeout -k 10 500 python bench.py "$@" --side-file "$OUT/$tag.full.json" > "$OUT/$tag.json" 2> "$OUT/$tag.err"
  rc=$?
  python - "$OUT/$tag.json" "$tag" $rc <<'P'
import json, sys
path, tag, rc = sys.argv[1:4]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(f"{tag}: rc {rc}, {len(open(path).read().strip().splitlines()[-1])} B, n_gpus {d['n_gpus']} ms/step {d['ms_per_step']:.2f} settings {d.get('settings')} alt {d.get('alt_settings', {}).get('ms_per_step')} secondary {d.get('secondary_ms')} errors {[k for k in d if 'error' in k or k == 'fallback']}")
except Exception as e:
    print(f"{tag}: rc {rc}, no line: {e}")
P
  [ $rc -ne 0 ] && tail -5 "$OUT/$tag.err"
}
run n2_memo --gpus 2 --steps 3 --warmup 1 --plan memo
run n2_reduce --gpus 2 --steps 3 --warmup 1 --exchange reduce --no-secondary
run n3_cfg2 --gpus 3 --steps 3 --warmup 1 --workload cfg2
run n2_cfg5 --gpus 2 --steps 2 --warmup 1 --workload cfg5
run n2_cfg4 --gpus 2 --steps 4 --warmup 2 --workload cfg4
run n2_sets --gpus 2 --steps 3 --warmup 1 --mode sets

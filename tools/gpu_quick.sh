#!/bin/bash
# fast loop: the GPU parity tests (optionally -k filter), then one cfg3 bench line
set -u
export PANO_BENCH_FULL_LINE=1   # the whole record on stdout (bench.py prints a compact line otherwise)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-q}
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -x -q ${2:+-k "$2"} > "$OUT/pytest.log" 2>&1; tail -4 "$OUT/pytest.log"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_cfg3.json" 2> "$OUT/bench.err"
python - "$OUT/bench_cfg3.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step %.3f" % d["ms_per_step"], {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()})
P

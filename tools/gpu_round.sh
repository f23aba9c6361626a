#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench lines, rocprofv3 kernel stats.
# Everything lands in gpurun_out/ (merged back into the repo by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r01}
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== device" > "$OUT/info.log"
rocm-smi --showproductname 2>&1 | head -20 >> "$OUT/info.log"
nproc >> "$OUT/info.log"
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee "$OUT/pytest_gpu.log"
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
echo "== bench"
timeout 600 python bench.py --workload cfg2 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | tee "$OUT/bench_cfg2.json"
timeout 900 python bench.py --steps 3 --warmup 1 2>&1 | tail -3 | tee "$OUT/bench_cfg3.json"
echo "== rocprof"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_cfg3" -- python3 "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OLDPWD/$OUT/rocprof.log" 2>&1
cd "$OLDPWD"
find "$OUT/prof_cfg3" -name "*kernel_stats*" | head -3
f=$(find "$OUT/prof_cfg3" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
# keep the merge small: drop the raw per-dispatch trace
find "$OUT/prof_cfg3" -name "*kernel_trace.csv" -size +20M -delete

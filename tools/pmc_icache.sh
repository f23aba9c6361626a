#!/bin/bash
# Instruction-fetch counters of the cfg3 bench (separate --pmc pass): tools/pmc_icache.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HERE=$PWD
OUT=$HERE/gpurun_out/${1:-pmc_icache}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQC_" | head -60 > "$OUT/counters_list.txt"
run() {
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$HERE/bench.py" --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary \
      > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run ic0 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run ic1 SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES
cd "$HERE"
python3 tools/pmc_summary.py "$OUT" cfg3 | tee "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +8M -delete

#!/bin/bash
# PMC passes over the cfg3 bench (separate rocprofv3 runs, --pmc only: no trace
# domains, see the pool rules).  tools/pmc.sh <tag> [workload] [by-grid]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HERE=$PWD
OUT=$HERE/gpurun_out/${1:-pmc}
WL=${2:-cfg3}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$HERE/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --busy-seconds 0 --side-file "$OUT/$name.full.json" ${PMC_EXTRA:-} \
      > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run sq0 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
cd "$HERE"
python3 tools/pmc_summary.py "$OUT" "$WL" ${3:-} | tee "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +8M -delete

#!/bin/bash
# memory-path counters of the matrix-core blur (separate --pmc passes).  tools/pmc_blur2.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HERE=$PWD
OUT=$HERE/gpurun_out/${1:-pmcb}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 "$HERE/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run ta1 TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run sqv1 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM
run sqv2 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run sqw SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
cd "$HERE"
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary_all.txt"
grep -A 40 "blur_mfma" "$OUT/summary_all.txt" | head -60 | tee "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +8M -delete

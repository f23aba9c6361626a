#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on this box (tools/probes/fetch_calib.hip, built here as
# build/fetch_calib):   tools/fetch_calib.sh <tag>   ->  gpurun_out/<tag>/fetch_calib.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HERE=$PWD
OUT=$HERE/gpurun_out/${1:-calib}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 120 "$HERE/build/fetch_calib" 1024 3 > "$OUT/known.json" 2> "$OUT/known.err" || { echo "fetch_calib failed"; cat "$OUT/known.err"; exit 1; }
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z]*REQ[A-Z0-9_]*" | sort -u > "$OUT/tcc_counters.txt"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  set -- $pass; name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- "$HERE/build/fetch_calib" 1024 1 > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
done
cd "$HERE"
python3 tools/fetch_calib_summary.py "$OUT" | tee "$OUT/fetch_calib.txt"

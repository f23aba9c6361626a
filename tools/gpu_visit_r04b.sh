#!/bin/bash
# Round 4, second GPU visit: calibration with the fixed byte / short kernels and the sized read
# requests, the blur's run statistics, counter passes of cfg3 / cfg4 with the rdreq pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-r04b}
OUT=gpurun_out/$T
mkdir -p "$OUT"
export TMPDIR=/tmp
tools/fetch_calib.sh "$T/calib"
echo "== run statistics"
timeout -k 10 300 python tools/probe_runs.py cfg3 > "$OUT/runs_cfg3.txt" 2>&1; cat "$OUT/runs_cfg3.txt"
timeout -k 10 300 python tools/probe_runs.py cfg2 > "$OUT/runs_cfg2.txt" 2>&1; cat "$OUT/runs_cfg2.txt"
timeout -k 10 600 python tools/probe_runs.py cfg5 > "$OUT/runs_cfg5.txt" 2>&1; cat "$OUT/runs_cfg5.txt"
echo "== PMC cfg3"
tools/pmc.sh "$T/pmc_cfg3" cfg3 > "$OUT/pmc_cfg3.log" 2>&1; grep -A3 "blur_lean\|multiband_compose\|warp_windows" "$OUT/pmc_cfg3/pmc_traffic.json" | head -40
echo "== PMC cfg4"
tools/pmc.sh "$T/pmc_cfg4" cfg4 > "$OUT/pmc_cfg4.log" 2>&1; tail -2 "$OUT/pmc_cfg4.log"

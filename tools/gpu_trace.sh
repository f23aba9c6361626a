#!/bin/bash
# kernel timeline of one cfg3 stitch (rocprofv3 --kernel-trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HERE=$PWD; OUT=$HERE/gpurun_out/${1:-trace}; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$HERE/bench.py" --workload ${2:-cfg3} --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/rocprof.log" 2>&1
cd "$HERE"
python3 - "$OUT" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/prof/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('ownership_cameras')]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:8.1f}  q{r['Queue_Id']} {r['Kernel_Name'][:44]}")
P
cp "$OUT"/prof/*/*kernel_stats.csv "$OUT/kernel_stats.csv"

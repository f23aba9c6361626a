#!/bin/bash
# bench lines only: N=1 and the 2-rank dry run (both ranks on one GPU, gloo rendezvous)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-b}
mkdir -p "$OUT"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err" ; tail -1 "$OUT/bench_cfg3.json" | cut -c1-600
PANO_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > "$OUT/bench_cfg3_2rank_dryrun.json" 2> "$OUT/bench_2rank.err"
tail -1 "$OUT/bench_cfg3_2rank_dryrun.json"; tail -5 "$OUT/bench_2rank.err"

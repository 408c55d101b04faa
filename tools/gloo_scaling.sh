#!/usr/bin/env bash
# Functional check of the sharded path on a ONE-GPU box: bench.py with 2 / 4 / 8 ranks, all
# on cuda:0, collectives through gloo (RCCL refuses several ranks per GPU).  The ranks
# time-share one GPU, so the rays/s are NOT a scaling measurement -- the lines show that the
# N-rank code path (voxel-balanced shards, all-reduce per BP iteration, all-gather of the
# depth rows) runs end to end at config-2 size, and what each rank launches.
# usage: bash tools/gloo_scaling.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
: > $OUT/${TAG}_gloo_ranks.jsonl
for N in 2 4 8; do
  RAYNET_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
    bench.py --gpus $N --steps 3 --warmup 1 --no-cpu-baseline 2>$OUT/${TAG}_gloo_$N.err \
    | tail -1 >> $OUT/${TAG}_gloo_ranks.jsonl
  echo "N=$N rc=$?"
done
cat $OUT/${TAG}_gloo_ranks.jsonl

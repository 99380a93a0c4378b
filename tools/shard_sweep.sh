#!/bin/bash
# cgs GMRES(100) through the sharded code path on one GPU (1-rank RCCL communicator) at the sizes one
# rank of 1 / 2 / 4 / 8 holds of the N = 10^7 problem:  bash tools/shard_sweep.sh
for ny in 2500 1250 625 313; do
  echo -n "ny=$ny: "
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --force-sharded --nx 4000 --ny $ny --no-cpu-baseline --no-roofline 2>/dev/null | cut -c80-110
done

#!/bin/bash
# Rehearsal of the data-parallel path on a ONE-GPU box: the exact launch line the multi-GPU bench
# uses, with one rank, forcing the RCCL rendezvous and the side-stream all-reduce of both gradient
# buffers (SIV_FORCE_COLLECTIVE=1).  The result must equal the single-process bench.
set -e
export SIV_FORCE_COLLECTIVE=1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline

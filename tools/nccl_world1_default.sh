#!/bin/bash
# Runs on the GPU box: bench.py's N > 1 headline with its DEFAULT sizes (shared batch 32) on a world of one rank,
# real backend (nccl = RCCL) and the library's communicator; side legs skipped.
R=$GRAFT_REPO_ROOT; cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
timeout -k 10 400 python bench.py --gpus 1 --steps 6 --warmup 2 --mode shared-factor --force-shared --no-side-legs

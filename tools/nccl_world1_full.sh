#!/bin/bash
# Runs on the GPU box: bench.py's complete N > 1 flow (headline + all side legs) with its default sizes on a world of one
# rank: real backend (nccl = RCCL), the library's communicator, the watchdog armed.
R=$GRAFT_REPO_ROOT; cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
timeout -k 10 800 python bench.py --gpus 1 --steps 5 --warmup 2 --mode shared-factor --force-shared

#!/bin/bash
# Runs on the GPU box: bench.py's N > 1 code with the REAL backend (nccl = RCCL) and the library's own
# communicator (gmrf_comm_*), on a world of ONE rank -- everything of the multi-GPU launch except a peer.
R=$GRAFT_REPO_ROOT; cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 --mode shared-factor --force-shared --batch 8 --streams 2 --shared-batch 2

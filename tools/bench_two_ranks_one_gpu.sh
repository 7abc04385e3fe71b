#!/bin/bash
# Runs on the GPU box: rehearse bench.py's N = 2 control flow (both modes) with two processes on ONE
# GPU over gloo -- the launch line is the driver's, only the backend differs.
R=$GRAFT_REPO_ROOT; cd $R
export GMRF_BENCH_BACKEND=gloo GMRF_BENCH_ONE_DEVICE=1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --streams 2 --no-cpu-baseline || exit 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 \
    bench.py --gpus 2 --steps 3 --warmup 1 --mode shared-factor --config darcy64 --no-cpu-baseline || exit 1

#!/bin/bash
# Runs on the GPU box: rehearse bench.py's N = 2 control flow with two processes on ONE GPU over gloo --
# the launch line is the driver's, only the backend differs (timings are meaningless: two ranks share
# one GPU and the factor travels through the host).  Default at N > 1: independent problems per rank (headline) + the
# shared-factor job, its regimes, the one-rank reference and C4 as side legs; then the shared factor as the headline.
R=$GRAFT_REPO_ROOT; cd $R
export GMRF_BENCH_BACKEND=gloo GMRF_BENCH_ONE_DEVICE=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps 4 --warmup 1 --batch 8 --streams 2 --shared-batch 2 --regimes 128 || exit 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 \
    bench.py --gpus 2 --steps 4 --warmup 1 --mode shared-factor --batch 8 --streams 2 --shared-batch 2 --regimes 128 || exit 1

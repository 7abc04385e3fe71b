#!/bin/bash
# Runs on the GPU box: rehearse bench.py's N = 2 control flow with two processes on ONE GPU over gloo --
# the launch line is the driver's, only the backend differs (timings are meaningless: two ranks share
# one GPU and the factor travels through the host).  Default mode at N > 1: shared factor (headline)
# + the one-rank reference, C4 and problems-mode legs; then the explicit problems mode.
R=$GRAFT_REPO_ROOT; cd $R
export GMRF_BENCH_BACKEND=gloo GMRF_BENCH_ONE_DEVICE=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps 3 --warmup 1 --batch 8 --streams 2 --shared-batch 2 || exit 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 \
    bench.py --gpus 2 --steps 3 --warmup 1 --mode problems --batch 8 --streams 2 || exit 1

#!/bin/bash
# Runs on the GPU box: A/B of set_eager flag sets on the SAME box (GPU boxes differ by > 10 %):
#   tools/ab_sweep.sh "<flagsA> <flagsB> ..." <streams>x<batch>
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  for fl in $1; do
    echo "flags=$fl"; GMRF_KEEP_L=0 GMRF_EAGER_FLAGS=$fl timeout -k 10 300 python tools/stream_sweep.py darcy256 ${2:-4x32} 2>&1 | grep solves
  done
done

#!/bin/bash
# Runs on the GPU box: K6 node-major SpMM, env switch A/B on the same box + the parity tests of the kernel
#   tools/spmm_ab.sh VAR "<valueA> <valueB> ..."
R=$GRAFT_REPO_ROOT; cd $R
for v in $2; do
  echo "== $1=$v"
  env $1=$v timeout -k 10 200 python tools/spmm_probe.py burgers4096x512 2>&1 | grep -E "rows|vector"
done

#!/bin/bash
# Runs on the GPU box: variants of the LDS-tiled SpMM kernel on one box (GMRF_SPMM_VARIANT x GMRF_SPMM_LDS_KB)
R=$GRAFT_REPO_ROOT; cd $R
for v in ${1:-0 1}; do for kb in ${2:-52 30}; do
  echo "variant $v lds budget $kb KB"; GMRF_SPMM_VARIANT=$v GMRF_SPMM_LDS_KB=$kb timeout -k 10 200 python tools/spmm_probe.py burgers4096x512 2>&1 | grep "fp64.*rows"
done; done

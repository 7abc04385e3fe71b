#!/bin/bash
# Round 5 A/B: one problem's small products on the 32 x 32-tile kernel (default) or on the LDS-DMA 64 x 64 kernel (GMRF_GEMM_LL_MAX_TILES=0)
set -o pipefail
OUT=gpurun_out/${1:-r5ll}; mkdir -p $OUT
for v in -1 0 40; do
  if [ $v -ge 0 ]; then export GMRF_GEMM_LL_MAX_TILES=$v; else unset GMRF_GEMM_LL_MAX_TILES; fi
  for cfg in darcy256 elliptic512; do
    timeout -k 10 300 python tools/probe.py $cfg 64 > $OUT/probe_${cfg}_$v.txt 2>&1 || { tail -20 $OUT/probe_${cfg}_$v.txt; exit 1; }
    echo "LL_MAX_TILES=$v $cfg: $(grep -E '^\[graph\]' $OUT/probe_${cfg}_$v.txt | tail -1)"
  done
done

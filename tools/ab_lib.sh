#!/bin/bash
# A/B of two builds of the library (GMRF_HIP_LIBRARY) and environment settings: GEMM rates, then the bench job.
OUT=gpurun_out/${1:-ab_lib}; mkdir -p $OUT
RF=$PWD/diffeqgmrfs.jl_amd/csrc/libgmrf_hip_rf.so
echo "== default build" | tee -a $OUT/rate.log; timeout -k 10 200 python tools/gemm_dma_rate.py 32 2>/dev/null | cut -c1-110 | tee -a $OUT/rate.log
echo "== reads-first build" | tee -a $OUT/rate.log; GMRF_HIP_LIBRARY=$RF timeout -k 10 200 python tools/gemm_dma_rate.py 32 2>/dev/null | cut -c1-110 | tee -a $OUT/rate.log
echo "== default build, 8 KB LDS pad (4 wgs/CU)" | tee -a $OUT/rate.log; GMRF_GEMM_DMA_LDS_PAD_KB=8 timeout -k 10 200 python tools/gemm_dma_rate.py 32 2>/dev/null | cut -c1-110 | tee -a $OUT/rate.log
echo "== default build, 21 KB LDS pad (3 wgs/CU)" | tee -a $OUT/rate.log; GMRF_GEMM_DMA_LDS_PAD_KB=21 timeout -k 10 200 python tools/gemm_dma_rate.py 32 2>/dev/null | cut -c1-110 | tee -a $OUT/rate.log
tools/ab_env2.sh $1 "X=0" "GMRF_HIP_LIBRARY=$RF" "GMRF_GEMM_DMA_LDS_PAD_KB=8" "X=0"

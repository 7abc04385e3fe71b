#!/bin/bash
# A/B of the GEMM staging (register-staged / LDS-DMA, stages) on the bench job within ONE gpurun call.
OUT=gpurun_out/${1:-ab_dma}; mkdir -p $OUT
for cfg in "0 2" "1 2" "1 3" "3 2" "1 2" "0 2"; do
  set -- $cfg
  GMRF_GEMM_DMA=$1 GMRF_GEMM_DMA_STAGES=$2 timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-spmm --no-single-problem --no-full-loop 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('dma=$1 stages=$2: %.0f solves/s  %.1f ms/step  gemm ms: %s' % (d['value'], d['ms_per_step'], {n: round(v['ms_per_step'],1) for n,v in k.items() if v['ms_per_step']>0 and 'gemm' in n}), d['roofline']['kernel'], round(d['roofline']['frac'],3))
" | tee -a $OUT/ab.log || exit 1
done

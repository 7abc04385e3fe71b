#!/bin/bash
# A/B of set_eager flag sets on the bench job within ONE gpurun call:  tools/ab_flags.sh <outdir> "<flags> <flags> ..."
OUT=gpurun_out/${1:-ab_flags}; mkdir -p $OUT
for fl in $2; do
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-spmm --no-single-problem --no-full-loop --eager-flags $fl 2>$OUT/err_$fl.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('flags=$fl: %.0f solves/s  %.1f ms/step  ms: %s' % (d['value'], d['ms_per_step'], {n: round(v['ms_per_step'],1) for n,v in k.items() if v['ms_per_step']>0}), d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline']['all_gemm_symbols_time_weighted']['frac'],3))
" | tee -a $OUT/ab.log || { tail -5 $OUT/err_$fl.log; exit 1; }
done

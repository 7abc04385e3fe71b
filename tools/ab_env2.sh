#!/bin/bash
# A/B of environment settings on the bench job within ONE gpurun call: tools/ab_env2.sh <outdir> "VAR=v,VAR2=w" "VAR=x" ...
OUT=gpurun_out/$1; mkdir -p $OUT; shift
for cfg in "$@"; do
  env $(echo $cfg | tr ',' ' ') timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-spmm --no-single-problem --no-full-loop 2>$OUT/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$cfg: %.0f solves/s  %.1f ms/step  ms: %s' % (d['value'], d['ms_per_step'], {n: round(v['ms_per_step'],1) for n,v in k.items() if v['ms_per_step']>0}))
" | tee -a $OUT/ab.log || { tail -5 $OUT/err.log; exit 1; }
done

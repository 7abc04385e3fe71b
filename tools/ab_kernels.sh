#!/bin/bash
# Runs on the GPU box: per-kernel-class times of one instrumented batch-32 step for several set_eager flag sets
R=$GRAFT_REPO_ROOT; cd $R
for fl in $1; do
  echo "== flags=$fl"
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-spmm --no-single-problem --no-full-loop --eager-flags $fl 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', round(d['value']), 'factor_ms', round(d['phases_ms']['factor'],2))
for k,v in d['kernels'].items():
    if v['ms_per_step']>0: print(f\"  {k:32s} {v['ms_per_step']:8.2f} ms  n={v['launches']:5d}  {list(v.values())[2]:.1f}\")
"
done

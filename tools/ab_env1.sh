#!/bin/bash
# Runs on the GPU box: per-kernel-class times (1 stream x batch 32) for values of one environment switch
R=$GRAFT_REPO_ROOT; cd $R
for v in $2; do
  echo "== $1=$v"
  env $1=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-spmm --no-single-problem --no-full-loop 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1x32 value', round(d['value']), 'phases', {k: round(v,2) for k,v in d['phases_ms'].items()})
print('  ', {k:(round(v['ms_per_step'],2), v['launches']) for k,v in d['kernels'].items() if v['ms_per_step']>0})
"
done

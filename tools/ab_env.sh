#!/bin/bash
# Runs on the GPU box: A/B of one environment switch on the SAME box, per-kernel-class times (1 stream x batch 32)
# and job throughput (4 x 32):   tools/ab_env.sh VAR "<valueA> <valueB> ..."
R=$GRAFT_REPO_ROOT; cd $R
VAR=$1
for v in $2; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-spmm --no-single-problem --no-full-loop 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1x32 value', round(d['value']), 'factor_ms', round(d['phases_ms']['factor'],2), 'gemm tw', round(d['roofline']['gemm_f64_mfma_both_symbols_time_weighted']['achieved'],1))
for k,v in d['kernels'].items():
    if v['ms_per_step']>0 and k.startswith('gemm'): print(f\"  {k:32s} {v['ms_per_step']:8.2f} ms  n={v['launches']:5d}  {list(v.values())[2]:.1f}\")
" || exit 1
done
for rep in 1 2; do
  for v in $2; do
    echo "$VAR=$v"; env $VAR=$v GMRF_KEEP_L=0 timeout -k 10 300 python tools/stream_sweep.py darcy256 4x32 2>&1 | grep solves
  done
done

#!/bin/bash
# throughput against the number of batched handles / batch size within ONE gpurun call: tools/ab_streams.sh <outdir> "SxB SxB ..."
OUT=gpurun_out/${1:-ab_streams}; mkdir -p $OUT
for sb in $2; do
  s=${sb%x*}; b=${sb#*x}
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --streams $s --batch $b --no-cpu-baseline --no-spmm --no-single-problem --no-full-loop 2>$OUT/err_$sb.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$sb: %.0f solves/s  %.1f ms/step  hbm %.0f GB  queues %s' % (d['value'], d['ms_per_step'], d['hbm_used_gb'], d.get('streams_on_own_hardware_queue')))
" | tee -a $OUT/ab.log || { tail -5 $OUT/err_$sb.log; exit 1; }
done

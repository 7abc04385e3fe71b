#!/bin/bash
# spmm_bxt_tiles by groups: chunks per workgroup
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4ab}; mkdir -p $OUT
cd $R
for v in 8 4 6 12 16 24; do
  ( cd /tmp && export TMPDIR=/tmp && export GMRF_BXT_NCH=$v && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-single-problem --no-spmm --no-full-loop > $OUT/bench_$v.json 2> $OUT/trace_$v.err ) || { tail -5 $OUT/trace_$v.err; exit 1; }
  python3 tools/trace_summary.py $(ls $OUT/trace_$v/*/*kernel_trace.csv | head -1) 1 45 > $OUT/by_grid_$v.txt
  rm -rf $OUT/trace_$v
  echo "nch cap $v: $(grep -i "spmm_bxt" $OUT/by_grid_$v.txt | tail -1)"
done

#!/bin/bash
# Runs on the GPU box: kernel trace of one stream x batch 16 (solo kernel durations) -> gpurun_out/solo/
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/solo; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --streams ${1:-1} --batch ${2:-16} --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.err || exit 1
python3 $R/tools/trace_summary.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 1 60 > $OUT/by_grid.txt
rm -rf $OUT/trace

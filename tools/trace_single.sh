#!/bin/bash
# Runs on the GPU box: kernel trace of ONE problem (tools/probe.py) summarised by kernel and grid.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/single; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/probe.py ${1:-darcy256} 64 > $OUT/probe.log 2> $OUT/err.log || exit 1
python3 $R/tools/trace_summary.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 9 70 > $OUT/by_grid.txt
rm -rf $OUT/trace

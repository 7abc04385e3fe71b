#!/bin/bash
# Runs on the GPU box: kernel durations of the SpMV / SpMM probe
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/spmm; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/spmm_probe.py > $OUT/probe.log 2> $OUT/err.log || exit 1
python3 $R/tools/trace_summary.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 1 30 > $OUT/by_grid.txt
rm -rf $OUT/trace

#!/bin/bash
# Round 5: rocprofv3 kernel traces of C4 (elliptic512, 1 stream x batch 8), C5 (burgers4096x512, one problem) and of one darcy256
# problem (tools/probe.py), each summarised by kernel symbol and grid.  $1 = output directory under gpurun_out/
set -o pipefail
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/${1:-r5traces}; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_c4 -- python3 $R/bench.py --config elliptic512 --batch 8 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $P/c4_1stream_under_rocprof.json 2> $P/trace_c4.err || { echo "c4 trace failed"; tail -5 $P/trace_c4.err; exit 1; }
python3 $R/tools/trace_summary.py $(ls $P/trace_c4/*/*kernel_trace.csv | head -1) 1 40 > $P/c4_trace_by_grid.txt
cp $(ls $P/trace_c4/*/*kernel_stats.csv | head -1) $P/c4_kernel_stats_1stream.csv
rm -rf $P/trace_c4
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_c5 -- python3 $R/bench.py --config burgers4096x512 --batch 1 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $P/c5_under_rocprof.json 2> $P/trace_c5.err || { echo "c5 trace failed"; tail -5 $P/trace_c5.err; exit 1; }
python3 $R/tools/trace_summary.py $(ls $P/trace_c5/*/*kernel_trace.csv | head -1) 1 40 > $P/c5_trace_by_grid.txt
cp $(ls $P/trace_c5/*/*kernel_stats.csv | head -1) $P/c5_kernel_stats.csv
rm -rf $P/trace_c5
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $P/trace_1 -- python3 $R/tools/probe.py darcy256 64 > $P/probe.log 2> $P/trace_1.err || { echo "single trace failed"; tail -5 $P/trace_1.err; exit 1; }
python3 $R/tools/trace_summary.py $(ls $P/trace_1/*/*kernel_trace.csv | head -1) 9 70 > $P/single_by_grid.txt
rm -rf $P/trace_1
head -28 $P/c5_trace_by_grid.txt; head -16 $P/c4_trace_by_grid.txt; head -16 $P/single_by_grid.txt

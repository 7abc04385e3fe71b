#!/bin/bash
# Runs on the GPU box: rocprofv3 summaries of the bench command -> gpurun_out/profiles/
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --batch 16 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --batch 16 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err || exit 1
python3 $R/tools/trace_summary.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 1 30 > $OUT/trace_by_grid.txt
ls -R $OUT | head -30

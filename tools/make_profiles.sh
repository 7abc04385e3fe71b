#!/bin/bash
# Runs on the GPU box: rocprofv3 summaries of the bench command -> gpurun_out/profiles/
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/profiles; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the default bench command (4 streams x batch 32: kernels of different streams overlap)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
# 2. one stream (no overlap: launch durations comparable with bench.py's instrumented roofline step)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-single-problem --no-spmm --no-full-loop > $OUT/bench_1stream_under_rocprof.json 2> $OUT/trace1.err || exit 1
# 3. HBM traffic counters, one pass each
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single-problem --no-spmm --no-full-loop > /dev/null 2> $OUT/pmc_fetch.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single-problem --no-spmm --no-full-loop > /dev/null 2> $OUT/pmc_write.err || exit 1
# 4. L2 hits / misses (one pass): what share of a kernel's L2 requests its own XCD's L2 served
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single-problem --no-spmm --no-full-loop > /dev/null 2> $OUT/pmc_tcc.err || exit 1
python3 - $(ls $OUT/pmc_tcc/*/*counter_collection.csv | head -1) > $OUT/pmc_l2_table.md <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gmrf::", "")
    a = agg[k]
    if r["Counter_Name"] == "TCC_HIT_sum": a[0] += 1; a[1] += float(r["Counter_Value"])
    elif r["Counter_Name"] == "TCC_MISS_sum": a[2] += float(r["Counter_Value"])
print("| kernel | launches | TCC_HIT_sum / launch | TCC_MISS_sum / launch | hit share |")
print("|---|---|---|---|---|")
for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    if a[0]: print(f"| {k} | {a[0]} | {a[1] / a[0]:.0f} | {a[2] / a[0]:.0f} | {a[1] / max(a[1] + a[2], 1.0):.3f} |")
PY
python3 $R/tools/trace_summary.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 1 45 > $OUT/trace_by_grid.txt
python3 $R/tools/trace_summary.py $(ls $OUT/trace1/*/*kernel_trace.csv | head -1) 1 45 > $OUT/trace1_by_grid.txt
python3 $R/tools/pmc_summary.py $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) $OUT/hbm_traffic.json > $OUT/pmc_table.md
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_4streams.csv
cp $(ls $OUT/trace1/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_1stream.csv
rm -rf $OUT/trace $OUT/trace1 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_tcc
ls -l $OUT

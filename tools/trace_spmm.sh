#!/bin/bash
# Runs on the GPU box: K6 on the burgers4096x512 precision matrix under rocprofv3 -> gpurun_out/spmm/
#   kernel durations (--kernel-trace --stats), then HBM traffic and L2 hit counters in separate --pmc passes
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/spmm; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/spmm_one.py > $OUT/one.log 2> $OUT/err.log || exit 1
python3 $R/tools/trace_summary.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 1 12 > $OUT/by_grid.txt
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/spmm_one.py > /dev/null 2> $OUT/pmc_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/spmm_one.py > /dev/null 2> $OUT/pmc_write.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $R/tools/spmm_one.py > /dev/null 2> $OUT/pmc_l2.err || exit 1
python3 $R/tools/pmc_summary.py $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) $OUT/hbm_traffic.json > $OUT/pmc_table.md
python3 - $(ls $OUT/pmc_l2/*/*counter_collection.csv | head -1) >> $OUT/pmc_table.md <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gmrf::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
print("\n| kernel | TCC_HIT_sum | TCC_MISS_sum | L2 hit rate |\n|---|---|---|---|")
for k, c in agg.items():
    h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    if h + m > 0: print(f"| {k} | {h:.3g} | {m:.3g} | {h / (h + m):.3f} |")
PY
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_l2
cat $OUT/by_grid.txt $OUT/pmc_table.md

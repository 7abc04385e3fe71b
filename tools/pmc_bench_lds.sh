#!/bin/bash
# Runs on the GPU box: LDS counters of every kernel of one bench step (1 stream x batch 32) -> gpurun_out/bench_lds.md
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bench_lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single-problem --no-spmm > /dev/null 2> $OUT/$tag.err || { echo "failed $tag"; tail -3 $OUT/$tag.err; exit 1; }
  cp $(ls $OUT/$tag/*/*counter_collection.csv | head -1) $OUT/$tag.csv
  rm -rf $OUT/$tag
done
python3 - $OUT <<'PY'
import csv, collections, sys, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/*.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gmrf::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
cols = ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_ADDR_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS"]
print("| kernel | " + " | ".join(cols) + " | conflict / active |"); print("|---|" + "---|" * (len(cols) + 1))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    a = c.get("SQ_LDS_IDX_ACTIVE", 0)
    print(f"| {k} | " + " | ".join(f"{c.get(x, 0):.3g}" for x in cols) + f" | {(c.get('SQ_LDS_BANK_CONFLICT', 0) / a if a else 0):.2f} |")
PY

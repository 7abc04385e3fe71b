#!/bin/bash
# Runs on the GPU box: SQ counters of every kernel of one bench step (1 stream x batch 32), one rocprofv3 --pmc pass per
# counter pair -> a table on stdout.   tools/pmc_bench_lds.sh [lds|issue]
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bench_lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "${1:-lds}" = "issue" ]; then
  SETS=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY")
  COLS='["SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"]'
  RATIO='("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "mfma busy / cu busy")'
else
  SETS=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS")
  COLS='["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_ADDR_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS"]'
  RATIO='("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "conflict / active")'
fi
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-single-problem --no-spmm --no-full-loop > /dev/null 2> $OUT/$tag.err || { echo "failed $tag"; tail -3 $OUT/$tag.err; exit 1; }
  cp $(ls $OUT/$tag/*/*counter_collection.csv | head -1) $OUT/$tag.csv
  rm -rf $OUT/$tag
done
python3 - $OUT "$COLS" "$RATIO" <<'PY'
import csv, collections, sys, glob, ast
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/*.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gmrf::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
cols = ast.literal_eval(sys.argv[2]); num, den, label = ast.literal_eval(sys.argv[3])
print("| kernel | " + " | ".join(cols) + f" | {label} |"); print("|---|" + "---|" * (len(cols) + 1))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    a = c.get(den, 0)
    print(f"| {k} | " + " | ".join(f"{c.get(x, 0):.3g}" for x in cols) + f" | {(c.get(num, 0) / a if a else 0):.2f} |")
PY

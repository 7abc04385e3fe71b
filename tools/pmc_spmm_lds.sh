#!/bin/bash
# Runs on the GPU box: LDS counters of the K6 kernels (tools/spmm_one.py) -> gpurun_out/spmm_lds/
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/spmm_lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/tools/spmm_one.py > /dev/null 2> $OUT/$tag.err || { echo "failed $tag"; tail -3 $OUT/$tag.err; continue; }
  python3 - $(ls $OUT/$tag/*/*counter_collection.csv | head -1) <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gmrf::", "")
    if not k.startswith("csr_sp"): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in agg.items():
    print(k, {a: f"{b / n[(k, a)]:.4g}" for a, b in c.items()})
PY
  rm -rf $OUT/$tag
done

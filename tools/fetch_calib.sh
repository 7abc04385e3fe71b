#!/bin/bash
# FETCH_SIZE by access width (tools/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- $R/tools/bin/fetch_calib > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/f/*/*counter_collection.csv")[0]
print("kernel | FETCH_SIZE (KiB) | as bytes / 2^30")
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        v = float(r["Counter_Value"])
        print(f'{r["Kernel_Name"][:60]:60s} | {v:14.0f} | {v * 1024 / 2**30:.3f}')
PY

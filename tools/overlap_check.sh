#!/bin/bash
# Runs on the GPU box: do the two branches of the captured factor graph overlap on the device?
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ovl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/probe.py darcy256 64 > $OUT/probe.log 2> $OUT/err.log || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = collections.Counter(r["Queue_Id"] for r in rows)
print("dispatches per queue:", dict(q))
# overlap: time covered by >= 2 kernels
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur = 0; last = ev[0][0]; busy1 = 0; busy2 = 0
for t, d in ev:
    if cur >= 1: busy1 += t - last
    if cur >= 2: busy2 += t - last
    cur += d; last = t
print(f"time with >=1 kernel {busy1/1e6:.1f} ms, with >=2 kernels {busy2/1e6:.1f} ms")
PY
rm -rf $OUT/trace

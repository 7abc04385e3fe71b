#!/bin/bash
# C4 with 7 workgroups per problem in the panel launches: batch 8 / 9 / 10 / 12 per handle at 4 streams
OUT=gpurun_out/${1:-r4x}; mkdir -p $OUT
run() {  # name, env, args
  env $2 timeout -k 10 400 python bench.py --config elliptic512 $3 --steps 3 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("$1", {k: d.get(k) for k in ("value", "ms_per_step", "hbm_used_gb")}, d.get("phases_ms"), d.get("batch_reduced"))
PY
}
run s4b8 "X=0" "--batch 8"
run s4b9 "X=0" "--batch 9"
run s4b10 "X=0" "--batch 10"
run s4b12 "X=0" "--batch 12"
run s3b12 "X=0" "--batch 12 --streams 3"

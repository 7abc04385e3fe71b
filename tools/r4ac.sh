#!/bin/bash
# headline: more hardware queues / more, smaller handles at the same number of posteriors in flight
OUT=gpurun_out/${1:-r4ac}; mkdir -p $OUT
run() {  # name, env, args
  env $2 timeout -k 10 400 python bench.py $3 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("$1", {k: d.get(k) for k in ("value", "ms_per_step", "hbm_used_gb", "streams_on_own_hardware_queue")})
PY
}
run s4b64 "X=0" ""
run q8_s4b64 "GPU_MAX_HW_QUEUES=8" ""
run q8_s8b32 "GPU_MAX_HW_QUEUES=8" "--streams 8 --batch 32"
run q8_s6b40 "GPU_MAX_HW_QUEUES=8" "--streams 6 --batch 40"
run q8_s5b48 "GPU_MAX_HW_QUEUES=8" "--streams 5 --batch 48"

#!/bin/bash
# C4: more hardware queues (GPU_MAX_HW_QUEUES) with more, smaller handles
OUT=gpurun_out/${1:-r4v}; mkdir -p $OUT
run() {  # name, env, args
  env $2 timeout -k 10 400 python bench.py --config elliptic512 $3 --steps 3 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("$1", {k: d.get(k) for k in ("value", "ms_per_step", "hbm_used_gb", "streams_on_own_hardware_queue")}, d.get("phases_ms"))
PY
}
run s4b8 "X=0" "--batch 8"
run q8_s4b8 "GPU_MAX_HW_QUEUES=8" "--batch 8"
run q8_s8b4 "GPU_MAX_HW_QUEUES=8" "--batch 4 --streams 8"
run q8_s6b5 "GPU_MAX_HW_QUEUES=8" "--batch 5 --streams 6"
run q8_s5b6 "GPU_MAX_HW_QUEUES=8" "--batch 6 --streams 5"

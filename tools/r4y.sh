#!/bin/bash
# C4 at 4 x 12: which launches the 32 x 32-tile GEMM kernel should take
OUT=gpurun_out/${1:-r4y}; mkdir -p $OUT
run() {  # name, env, args
  env $2 timeout -k 10 400 python bench.py --config elliptic512 $3 --steps 3 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("$1", {k: d.get(k) for k in ("value", "ms_per_step", "hbm_used_gb")}, d.get("phases_ms"), d.get("batch_reduced"))
PY
}
run s4b12 "X=0" "--batch 12"
run s4b12_ll192 "GMRF_GEMM_LL_MAX_TILES=192" "--batch 12"
run s4b12_ll384 "GMRF_GEMM_LL_MAX_TILES=384" "--batch 12"
run s4b8_ll64 "GMRF_GEMM_LL_MAX_TILES=64" "--batch 8"
run s4b8_ll256 "GMRF_GEMM_LL_MAX_TILES=256" "--batch 8"

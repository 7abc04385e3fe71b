#!/bin/bash
# C4: wide GEMM tiles (128 x 64) wherever they fit, single-stage policy
OUT=gpurun_out/${1:-r4aa}; mkdir -p $OUT
run() {  # name, env, args
  env $2 timeout -k 10 400 python bench.py --config elliptic512 $3 --steps 3 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("$1", {k: d.get(k) for k in ("value", "ms_per_step")}, d.get("phases_ms"), "tw", (r.get("all_gemm_symbols_time_weighted") or {}).get("achieved"))
for g in sorted(d["gemm_by_shape"], key=lambda g: -g["ms"])[:6]: print("    ", g["kernel"], g["MxNxK"], g["tri"], g["lower"], g["kb"], round(g["ms"], 2), round(g["tflops"], 1))
PY
}
run s4b8 "X=0" "--batch 8"
run s4b8_wide "GMRF_GEMM_DMA=3" "--batch 8"
run s4b8_st3 "GMRF_GEMM_DMA_STAGES=3" "--batch 8"

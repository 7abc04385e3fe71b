#!/bin/bash
# self-cleaning flag words (no memset node per persistent launch) + 32 x 32 GEMM kernel for batches of eight: tests of the
# persistent routes, then the lines that should move
set -o pipefail
OUT=gpurun_out/${1:-r4p}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "persist or panel or inverse_rows or config_ or measured or split or rehears or packed or potrf" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log | cut -c1-300; exit 1; }
tail -2 $OUT/pytest.log
line() {  # name env args
  env $2 timeout -k 10 400 python bench.py $3 --no-cpu-baseline --no-spmm --no-full-loop > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("$1", {k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), "tw", (r.get("all_gemm_symbols_time_weighted") or {}).get("frac"), (r.get("all_gemm_symbols_time_weighted") or {}).get("symbols"))
PY
}
line darcy256 "X=0" "" &&
line darcy256_ll256 "GMRF_GEMM_LL_MAX_TILES=256" "--no-single-problem" &&
line elliptic512 "X=0" "--config elliptic512 --batch 8 --steps 3 --warmup 1" &&
line burgers4096 "X=0" "--config burgers4096x512 --batch 1 --streams 1 --steps 2 --warmup 1" &&
line burgers512 "X=0" "--config burgers512x64"

#!/bin/bash
# G2 / staircase launches of the triangular grid: rows ascending (longest K first) against the old descending order
set -o pipefail
OUT=gpurun_out/${1:-r4t}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "gemm or staircase or measured or exact" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log | cut -c1-300; exit 1; }
tail -2 $OUT/pytest.log
line() {  # name env
  env $2 timeout -k 10 400 python bench.py --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
g2 = [g for g in d["gemm_by_shape"] if g["MxNxK"] == [768, 768, 768] and g["kb"] == 1]
print("$1", {k: d.get(k) for k in ("value", "ms_per_step")}, "tw", (r.get("all_gemm_symbols_time_weighted") or {}).get("frac"), "roofline", r.get("frac"), "G2", [(round(g["ms"], 2), round(g["tflops"], 1)) for g in g2], d.get("phases_ms"))
PY
}
line asc "X=0" && line desc "GMRF_GEMM_G2_ORDER=0" && line asc2 "X=0" && line desc2 "GMRF_GEMM_G2_ORDER=0"

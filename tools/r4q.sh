#!/bin/bash
# gemm_f64_dma with A stored [k][m]; selected inversion without its transposing passes
set -o pipefail
OUT=gpurun_out/${1:-r4q}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "gemm or var or exact or selected or marginal or measured" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log | cut -c1-300; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 300 python tools/var_profile.py 64 > $OUT/var_profile.log 2> $OUT/var_profile.err || { tail -5 $OUT/var_profile.err; exit 1; }
cat $OUT/var_profile.log
timeout -k 10 500 python bench.py --no-cpu-baseline --no-spmm --no-single-problem > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, "full_loop", (d.get("full_loop") or {}).get("ms_per_problem"))
PY

#!/bin/bash
# variance tests with out= on the device; default bench line (full loop with device-resident variances); C4: how many persistent
# workgroups (9 per problem, one per CU) the streams may hold together
set -o pipefail
OUT=gpurun_out/${1:-r4r}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "var or exact or marginal" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log | cut -c1-300; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 500 python bench.py --no-cpu-baseline --no-spmm --no-single-problem > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, "full_loop", (d.get("full_loop") or {}).get("ms_per_problem"))
PY
run() {  # name, env, args
  env $2 timeout -k 10 400 python bench.py --config elliptic512 $3 --steps 3 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("$1", {k: d.get(k) for k in ("value", "ms_per_step", "hbm_used_gb")}, d.get("phases_ms"))
PY
}
run s4b8 "X=0" "--batch 8"
run s4b7 "X=0" "--batch 7"
run s4b6 "X=0" "--batch 6"
run s3b9 "X=0" "--batch 9 --streams 3"
run s2b14 "X=0" "--batch 14 --streams 2"
run s4b8_nopp "GMRF_PERSIST_PANELS=0" "--batch 8"

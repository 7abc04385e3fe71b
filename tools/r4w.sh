#!/bin/bash
# 8 workgroups per problem in the persistent launches with inverse rows (the diagonal tiles' workgroups take the column-0
# inverse tiles): parity, then the lines that could move
set -o pipefail
OUT=gpurun_out/${1:-r4w}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "persist or panel or inverse_rows or abort or config_ or measured or split or rehears or packed or potrf" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log | cut -c1-300; exit 1; }
tail -2 $OUT/pytest.log
line() {  # name env args
  env $2 timeout -k 10 400 python bench.py $3 --no-cpu-baseline --no-spmm --no-full-loop > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("$1", {k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), d.get("phases_ms"))
PY
}
line darcy256 "X=0" "" &&
line elliptic512 "X=0" "--config elliptic512 --batch 8 --steps 3 --warmup 1" &&
line elliptic512_b "X=0" "--config elliptic512 --batch 8 --steps 3 --warmup 1 --no-single-problem" &&
line burgers512 "X=0" "--config burgers512x64"

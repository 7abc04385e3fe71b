#!/bin/bash
# tile Cholesky: the factored panel is copied into the tile by wave 3 beside the sub-tile updates (not by wave 0 at the end of
# its panel): parity of every potrf kernel, chain stamps, lines
set -o pipefail
OUT=gpurun_out/${1:-r4z}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "potrf or persist or panel or inverse_rows or abort or config_ or measured or split or golden or tile" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log | cut -c1-300; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
sed -n 4,7p $OUT/stamps.log; tail -1 $OUT/stamps.log
timeout -k 10 120 python tools/tile_timing.py > $OUT/tile.log 2>&1; tail -12 $OUT/tile.log
line() {  # name env args
  env $2 timeout -k 10 400 python bench.py $3 --no-cpu-baseline --no-spmm --no-full-loop > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("$1", {k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), d.get("phases_ms"), (d.get("kernels") or {}).get("potrf_diag128"))
PY
}
line darcy256 "X=0" "" &&
line elliptic512 "X=0" "--config elliptic512 --batch 8 --steps 3 --warmup 1 --no-single-problem" &&
line burgers4096 "X=0" "--config burgers4096x512 --batch 1 --streams 1 --steps 2 --warmup 1"

#!/bin/bash
# spmm_bxt_tiles with two chunks of X in flight: parity tests, its duration in a 1-stream trace, the headline
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4u}; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "bxt or staircase or darcy256 or measured or sparse or config_ or coupling" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log | cut -c1-300; exit 1; }
tail -2 $OUT/pytest.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-single-problem --no-spmm --no-full-loop > $OUT/bench_1stream_under_rocprof.json 2> $OUT/trace1.err ) || { tail -5 $OUT/trace1.err; exit 1; }
python3 tools/trace_summary.py $(ls $OUT/trace1/*/*kernel_trace.csv | head -1) 1 45 > $OUT/trace1_by_grid.txt
rm -rf $OUT/trace1
grep -i "spmm_bxt" $OUT/trace1_by_grid.txt
timeout -k 10 400 python bench.py --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, d["kernels"].get("spmm_bxt"), d.get("phases_ms"))
PY

#!/bin/bash
# Round 4: stamps + the whole GPU suite + one-problem probe (persistent / launch per step) + the default bench line.
set -o pipefail
OUT=gpurun_out/${1:-r4c}; mkdir -p $OUT
timeout -k 10 120 python tools/tile_timing.py > $OUT/tile.log 2>&1 || exit 1
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
echo "== persistent" > $OUT/probe.log; timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep -v "^profile" >> $OUT/probe.log || exit 1
echo "== launch per step" >> $OUT/probe.log; GMRF_PERSIST=0 timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep -v "^profile" >> $OUT/probe.log || exit 1
cat $OUT/probe.log | grep -E "==|graph"
if [ "$2" != "notest" ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
fi
if [ "$3" != "nobench" ]; then
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), "gemm tw", d.get("roofline", {}).get("all_gemm_symbols_time_weighted"))
PY
fi

#!/bin/bash
# Round 5: tile timing, chain stamps, then the GPU suite ($2 = a pytest -k expression, empty: everything), then (if $3 = bench) the default bench line.
set -o pipefail
OUT=gpurun_out/${1:-r5suite}; mkdir -p $OUT
timeout -k 10 120 python tools/tile_timing.py > $OUT/tile.txt 2>&1 || { cat $OUT/tile.txt; exit 1; }
grep -v amdgpu.ids $OUT/tile.txt | head -8
timeout -k 10 120 python tools/persist_stamps.py > $OUT/persist_stamps.txt 2>&1 || { cat $OUT/persist_stamps.txt; exit 1; }
grep -v amdgpu.ids $OUT/persist_stamps.txt | tail -19
if [ -n "$2" ]; then KEXPR=(-k "$2"); else KEXPR=(); fi
timeout -k 10 1100 python -m pytest tests -x -q -m gpu "${KEXPR[@]}" > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log | cut -c1-400; exit 1; }
tail -3 $OUT/pytest.log
if [ "$3" = "bench" ]; then
  timeout -k 10 500 python bench.py > $OUT/bench_line_default.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/bench_line_default.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print({k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), "roofline frac", r.get("frac"), "tw", r.get("all_gemm_symbols_time_weighted", {}).get("frac"))
PY
fi

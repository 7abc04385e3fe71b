#!/bin/bash
# Round 5: the default bench line (no profiler) and the other BASELINE configs through the same bench.py.  $1 = output directory
set -o pipefail
OUT=gpurun_out/${1:-r5bench}; mkdir -p $OUT
timeout -k 10 560 python bench.py > $OUT/bench_line_default.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$OUT/bench_line_default.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print({k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), "roofline frac", r.get("frac"), "tw", r.get("all_gemm_symbols_time_weighted", {}).get("frac"))
print("phases", d.get("phases_ms"))
print("full_loop", (d.get("full_loop") or {}).get("ms_per_problem"))
print("cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
PY
rm -f $OUT/bench_lines_other_configs.jsonl
for cfg in "burgers512x64" "darcy64" "elliptic512 --batch 8 --steps 3 --warmup 1" "burgers4096x512 --batch 1 --streams 1 --steps 2 --warmup 1"; do
  name=$(echo $cfg | cut -d' ' -f1)
  timeout -k 10 500 python bench.py --config $cfg --no-cpu-baseline --no-spmm --no-full-loop > $OUT/tmp.json 2> $OUT/bench_$name.err || { tail -20 $OUT/bench_$name.err; exit 1; }
  tail -1 $OUT/tmp.json >> $OUT/bench_lines_other_configs.jsonl
  python - <<PY
import json
d = json.loads(open("$OUT/tmp.json").read().strip().splitlines()[-1])
print("$name", {k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), "phases", d.get("phases_ms"))
PY
done
rm -f $OUT/tmp.json

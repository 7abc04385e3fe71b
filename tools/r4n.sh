#!/bin/bash
OUT=gpurun_out/${1:-r4n}; mkdir -p $OUT
timeout -k 10 300 python tools/rehearse_driver.py $OUT/reh > $OUT/reh.log 2>&1; echo "rehearsal rc $?"; tail -3 $OUT/reh.log | cut -c1-600
grep -v "amdgpu.ids" $OUT/reh/two_ranks.log | grep -n "Error\|error\|Traceback\|rank0\]" | head -20
for env in "GMRF_PERSIST_PANELS=1" "GMRF_PERSIST_PANELS=0"; do
  env $env timeout -k 10 500 python bench.py --config elliptic512 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/bench_c4_$env.json 2> $OUT/bench_c4_$env.err || { tail -20 $OUT/bench_c4_$env.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/bench_c4_$env.json").read().strip().splitlines()[-1])
print("elliptic512 4x8 $env", {k: d.get(k) for k in ("value", "ms_per_step")}, d.get("phases_ms"))
print("   kernels", {k: (round(v["ms_per_step"], 2), v["launches"]) for k, v in d.get("kernels", {}).items() if v["launches"]})
PY
done

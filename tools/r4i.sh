#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r4i}; mkdir -p $OUT
timeout -k 10 120 python tools/tile_timing.py 2>&1 | grep -E "err L|tile kernel|assembly|pf3 end|kernel end" 
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
sed -n 3,7p $OUT/stamps.log; tail -1 $OUT/stamps.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "potrf or inverse_rows or eager_and_graph or varianc or split_inverse or two_level or batch or measured_path or staircase or left_looking or without_the_l or large_block or burgers4096 or golden or factor_blocks" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep graph | tail -1
timeout -k 10 300 python tools/var_profile.py 64 2>&1 | grep -v amdgpu.ids | head -2
for env in "GMRF_GEMM128=0" "GMRF_GEMM128=1"; do
  env $env timeout -k 10 400 python bench.py --no-cpu-baseline --no-spmm --no-full-loop --steps 12 > $OUT/bench_$env.json 2> $OUT/bench_$env.err || { tail -20 $OUT/bench_$env.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$env.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("$env", {k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), "gemm tw", r.get("all_gemm_symbols_time_weighted", {}).get("frac"), r.get("all_gemm_symbols_time_weighted", {}).get("ms_per_step"), "phases", d.get("phases_ms"))
print("   kernels", {k: (round(v["ms_per_step"], 2), v["launches"]) for k, v in d.get("kernels", {}).items() if v["launches"]})
PY
done
for cfg in "burgers4096x512 --batch 1 --streams 1 --steps 2 --warmup 1" "elliptic512 --batch 8 --steps 3 --warmup 1"; do
  name=$(echo $cfg | cut -d' ' -f1)
  timeout -k 10 500 python bench.py --config $cfg --no-cpu-baseline --no-spmm --no-full-loop > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -20 $OUT/bench_$name.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1])
print("$name", {k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), d.get("phases_ms"))
PY
done

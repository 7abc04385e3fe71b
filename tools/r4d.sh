#!/bin/bash
# Round 4: chain with prefetch; C4 / C5 lines
set -o pipefail
OUT=gpurun_out/${1:-r4d}; mkdir -p $OUT
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
sed -n 3,6p $OUT/stamps.log; tail -1 $OUT/stamps.log
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "potrf or inverse_rows or eager_and_graph or factor_blocks_match or large_block_size or degenerate or ragged or mean_and_half or burgers4096" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
echo "== persistent" > $OUT/probe.log; timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep -v "^profile" >> $OUT/probe.log || exit 1
grep graph $OUT/probe.log | tail -1
for cfg in "burgers4096x512 --batch 1 --streams 1 --steps 2 --warmup 1" "elliptic512 --batch 8 --steps 3 --warmup 1" "burgers512x64 --steps 5" ; do
  name=$(echo $cfg | cut -d' ' -f1)
  timeout -k 10 500 python bench.py --config $cfg --no-cpu-baseline --no-spmm --no-full-loop > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -20 $OUT/bench_$name.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("$OUT/bench_$name.json").read().strip().splitlines()[-1])
print("$name", {k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"), "one handle", d.get("one_handle_phase_ms") or d.get("per_handle_ms"))
PY
done

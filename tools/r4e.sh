#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r4e}; mkdir -p $OUT
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
sed -n 3,8p $OUT/stamps.log; tail -1 $OUT/stamps.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "potrf or inverse_rows or eager_and_graph or varianc or split_inverse or measured_path or packed_transport or known_answer or sample_cov or without_the_l or config_darcy256" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
echo "== persistent" > $OUT/probe.log; timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep -v "^profile" >> $OUT/probe.log || exit 1
grep graph $OUT/probe.log | tail -1
timeout -k 10 500 python bench.py --no-cpu-baseline --no-spmm > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"))
print("full_loop", json.dumps(d.get("full_loop"))[:1500])
PY

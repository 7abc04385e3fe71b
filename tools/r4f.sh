#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r4f}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 500 python bench.py --no-spmm > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, "single", d.get("single_problem"))
print("full_loop", json.dumps(d.get("full_loop"))[:1200])
print("cpu", json.dumps(d.get("cpu_baseline"))[:900])
PY
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 --force-shared --batch 8 --streams 2 --shared-batch 2 --no-cpu-baseline --regimes 128 > $OUT/world1.json 2> $OUT/world1.err || { tail -20 $OUT/world1.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$OUT/world1.json").read().strip().splitlines()[-1])
print("world1", {k: d.get(k) for k in ("value", "shared_factor_solves_per_s", "shared_factor_form", "shared_factor_broadcast_solves_per_s", "bytes_per_link_per_step", "rccl_ranks", "c4_elliptic512")})
print("side error", (d.get("side_legs") or {}).get("error"))
PY

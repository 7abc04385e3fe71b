#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r4k}; mkdir -p $OUT
timeout -k 10 120 python tools/tile_timing.py 2>&1 | grep -E "err L|tile kernel|assembly|pf3 end|kernel end"
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
sed -n 3,9p $OUT/stamps.log; tail -1 $OUT/stamps.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "potrf or inverse_rows or eager_and_graph or two_level or large_block or golden or factor_blocks or degenerate or ragged or batch_of" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep graph | tail -1

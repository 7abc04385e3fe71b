#!/bin/bash
set -o pipefail
OUT=gpurun_out/${1:-r4g}; mkdir -p $OUT
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
sed -n 3,8p $OUT/stamps.log; tail -1 $OUT/stamps.log
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "potrf or inverse_rows or eager_and_graph or varianc or split_inverse or large_block or ragged or degenerate" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep graph | tail -1
timeout -k 10 300 python tools/var_profile.py 64 > $OUT/var_profile.log 2>&1 || { tail -20 $OUT/var_profile.log; exit 1; }
cat $OUT/var_profile.log
timeout -k 10 500 python tools/clock_probe.py 64 > $OUT/clock.log 2>&1 || { tail -20 $OUT/clock.log; exit 1; }
cat $OUT/clock.log

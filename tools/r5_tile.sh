#!/bin/bash
# Round 5: the streaming tile Cholesky alone , then the kernel tests and the parity tests named in $2.
set -o pipefail
OUT=gpurun_out/${1:-r5tile}; mkdir -p $OUT
timeout -k 10 120 python tools/tile_timing.py > $OUT/tile4.txt 2>&1 || { cat $OUT/tile4.txt; exit 1; }
grep -v amdgpu.ids $OUT/tile4.txt
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $OUT/pytest_kernels.log 2>&1 || { tail -40 $OUT/pytest_kernels.log | cut -c1-300; exit 1; }
tail -3 $OUT/pytest_kernels.log
if [ -n "$2" ]; then
  timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" > $OUT/pytest_parity.log 2>&1 || { tail -60 $OUT/pytest_parity.log | cut -c1-400; exit 1; }
  tail -3 $OUT/pytest_parity.log
fi

#!/bin/bash
# Round 4, first GPU run: persistent in-block Cholesky (one problem) -- parity, chain stamps, A/B against the launch-per-step form.
set -o pipefail
OUT=gpurun_out/${1:-r4a}; mkdir -p $OUT
FLAT=$PWD/diffeqgmrfs.jl_amd/csrc/libgmrf_hip_flat.so
echo "== tile timing, ds (default)" > $OUT/tile.log; timeout -k 10 120 python tools/tile_timing.py >> $OUT/tile.log 2>&1 || exit 1
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
timeout -k 10 120 python tools/persist_stamps.py 256 >> $OUT/stamps.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "potrf or inverse_rows or eager_and_graph or factor_blocks_match or large_block_size or degenerate or ragged or mean_and_half" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
echo "== persistent" > $OUT/probe.log; timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep -v "^profile" >> $OUT/probe.log || exit 1
echo "== launch per step" >> $OUT/probe.log; GMRF_PERSIST=0 timeout -k 10 200 python tools/probe.py darcy256 64 2>&1 | grep -v "^profile" >> $OUT/probe.log || exit 1
cat $OUT/probe.log

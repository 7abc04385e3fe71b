#!/bin/bash
# Round 5: the whole GPU suite, chain stamps, tile timing, and the real one-problem times (probe.py) -- one gpurun call.  $1 = output directory
set -o pipefail
OUT=gpurun_out/${1:-r5full}; mkdir -p $OUT
timeout -k 10 120 python tools/tile_timing.py > $OUT/tile.txt 2>&1 || { cat $OUT/tile.txt; exit 1; }
timeout -k 10 120 python tools/persist_stamps.py > $OUT/persist_stamps.txt 2>&1 || { cat $OUT/persist_stamps.txt; exit 1; }
grep -v amdgpu.ids $OUT/persist_stamps.txt | tail -4
for cfg in darcy256 elliptic512; do
  timeout -k 10 400 python tools/probe.py $cfg 64 > $OUT/probe_$cfg.txt 2>&1 || { tail -20 $OUT/probe_$cfg.txt; exit 1; }
  echo "$cfg: $(grep -E '^\[graph\]' $OUT/probe_$cfg.txt | tail -1)"
done
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log | cut -c1-400; exit 1; }
tail -14 $OUT/pytest.log

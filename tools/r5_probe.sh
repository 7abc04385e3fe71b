#!/bin/bash
# Round 5: chain stamps + the real times of one problem (probe.py) for the configs in $2..; $1 = output directory
set -o pipefail
OUT=gpurun_out/${1:-r5probe}; mkdir -p $OUT; shift
timeout -k 10 120 python tools/persist_stamps.py > $OUT/persist_stamps.txt 2>&1 || { cat $OUT/persist_stamps.txt; exit 1; }
grep -v amdgpu.ids $OUT/persist_stamps.txt | tail -6
for cfg in "${@:-darcy256}"; do
  timeout -k 10 400 python tools/probe.py $cfg 64 > $OUT/probe_$cfg.txt 2>&1 || { tail -20 $OUT/probe_$cfg.txt; exit 1; }
  grep -E "^\[graph\]|residual" $OUT/probe_$cfg.txt | tail -3
done

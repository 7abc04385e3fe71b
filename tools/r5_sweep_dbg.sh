#!/bin/bash
# Round 5 (tuning aid): the persistent k = 1 sweeps with their matrix loads skipped (GMRF_SWEEP_DBG=1, garbage results): what the hand-off alone costs
set -o pipefail
OUT=gpurun_out/${1:-r5swdbg}; mkdir -p $OUT
for d in 0 1; do
  GMRF_SWEEP_DBG=$d timeout -k 10 120 python tools/sweep_persist_check.py darcy256 > $OUT/sweep_check_dbg$d.txt 2>&1 || { tail -30 $OUT/sweep_check_dbg$d.txt; exit 1; }
  echo "GMRF_SWEEP_DBG=$d: $(grep '\[persist\] forward' $OUT/sweep_check_dbg$d.txt)"
done

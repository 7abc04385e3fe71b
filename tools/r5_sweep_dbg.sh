#!/bin/bash
# Round 5 (tuning aid): the persistent k = 1 sweeps with their matrix loads skipped (GMRF_SWEEP_DBG=1, garbage results): what the hand-off alone costs
set -o pipefail
OUT=gpurun_out/${1:-r5swdbg}; mkdir -p $OUT; shift
GMRF_SWEEP_DBG=1 timeout -k 10 300 python tools/sweep_persist_check.py "$@" > $OUT/sweep_check_dbg.txt 2>&1 || { tail -30 $OUT/sweep_check_dbg.txt; exit 1; }
grep "\[persist\]" $OUT/sweep_check_dbg.txt

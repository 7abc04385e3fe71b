#!/bin/bash
# Round 5: the persistent sweep against a launch per product (bitwise + times); $1 = output directory, $2.. = configs
set -o pipefail
OUT=gpurun_out/${1:-r5sweep}; mkdir -p $OUT; shift
timeout -k 10 900 python tools/sweep_persist_check.py "$@" > $OUT/sweep_check.txt 2>&1 || { tail -30 $OUT/sweep_check.txt; exit 1; }
grep -v amdgpu.ids $OUT/sweep_check.txt | tail -12

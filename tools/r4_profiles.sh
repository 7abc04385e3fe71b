#!/bin/bash
# Round 4: every rocprofv3 / probe summary that profiles/r04_* holds, one gpurun call
set -o pipefail
R=$GRAFT_REPO_ROOT
bash $R/tools/make_profiles.sh || { echo "make_profiles failed"; exit 1; }
bash $R/tools/trace_single.sh darcy256 || { echo "trace_single failed"; exit 1; }
cd $R
OUT=gpurun_out/r4prof; mkdir -p $OUT
timeout -k 10 120 python tools/persist_stamps.py 1024 > $OUT/stamps.log 2>&1 || { tail -5 $OUT/stamps.log; exit 1; }
timeout -k 10 300 python tools/var_profile.py 64 > $OUT/var_profile.log 2> $OUT/var_profile.err || { tail -20 $OUT/var_profile.err; exit 1; }
timeout -k 10 500 python tools/clock_probe.py 64 > $OUT/clock.log 2>&1 || { tail -20 $OUT/clock.log; exit 1; }
tail -3 $OUT/var_profile.log; tail -5 $OUT/clock.log; head -20 gpurun_out/single/by_grid.txt

#!/bin/bash
# Round 5: everything profiles/r05_* holds beside the bench lines, one gpurun call: rocprofv3 passes of the bench command (make_profiles.sh),
# kernel traces of the single problem / C4 / C5 (r5_traces.sh), the persistent sweep against a launch per product, its hand-off alone.
set -o pipefail
R=$GRAFT_REPO_ROOT
bash $R/tools/make_profiles.sh || { echo "make_profiles failed"; exit 1; }
bash $R/tools/r5_traces.sh r5traces > $R/gpurun_out/r5traces.log 2>&1 || { echo "r5_traces failed"; tail -20 $R/gpurun_out/r5traces.log; exit 1; }
cd $R
bash tools/r5_sweep.sh r5sweepF burgers512x64 darcy256 elliptic512 || { echo "sweep check failed"; exit 1; }
bash tools/r5_sweep_dbg.sh r5sweepF || exit 1

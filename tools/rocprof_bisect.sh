#!/bin/bash
# Runs on the GPU box: which bench variant survives rocprofv3 --kernel-trace (diagnosis of a tool-side crash)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bisect; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-spmm "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$? $(grep -c SIGSEGV $OUT/$name.err)"
  rm -rf $OUT/$name
}
run s4 --streams 4 --no-single-problem
run s1_single --streams 1
run s4_single --streams 4

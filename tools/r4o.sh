#!/bin/bash
# C4 (elliptic512) stream x batch / small-GEMM-kernel variants, one gpurun call
OUT=gpurun_out/${1:-r4o}; mkdir -p $OUT
run() {  # name, env, args
  env $2 timeout -k 10 400 python bench.py --config elliptic512 $3 --steps 3 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $OUT/$1.json 2> $OUT/$1.err || { echo "$1 failed"; tail -5 $OUT/$1.err; return; }
  python - <<PY
import json
d = json.loads(open("$OUT/$1.json").read().strip().splitlines()[-1])
print("$1", {k: d.get(k) for k in ("value", "ms_per_step", "hbm_used_gb")}, d.get("phases_ms"), d.get("batch_reduced"))
PY
}
run s4b8 "X=0" "--batch 8"
run s4b8_ll "GMRF_GEMM_LL_MAX_TILES=128" "--batch 8"
run s2b16 "X=0" "--batch 16 --streams 2"
run s3b8 "X=0" "--batch 8 --streams 3"
run s1b24 "X=0" "--batch 24 --streams 1"
run s5b8 "X=0" "--batch 8 --streams 5"
# kernel trace of C4 at 1 stream x batch 8 (durations without overlap) -> by-kernel / by-grid table
R=$GRAFT_REPO_ROOT; P=$R/$OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_c4 -- python3 $R/bench.py --config elliptic512 --batch 8 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $P/c4_1stream_under_rocprof.json 2> $P/trace_c4.err ) || { echo "c4 trace failed"; tail -5 $P/trace_c4.err; exit 1; }
python3 $R/tools/trace_summary.py $(ls $P/trace_c4/*/*kernel_trace.csv | head -1) 1 40 > $P/c4_trace_by_grid.txt
cp $(ls $P/trace_c4/*/*kernel_stats.csv | head -1) $P/c4_kernel_stats_1stream.csv
rm -rf $P/trace_c4
head -50 $P/c4_trace_by_grid.txt
# kernel trace of C5 (one problem, 512 blocks of 4096)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_c5 -- python3 $R/bench.py --config burgers4096x512 --batch 1 --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-spmm --no-full-loop --no-single-problem > $P/c5_under_rocprof.json 2> $P/trace_c5.err ) || { echo "c5 trace failed"; tail -5 $P/trace_c5.err; exit 1; }
python3 $R/tools/trace_summary.py $(ls $P/trace_c5/*/*kernel_trace.csv | head -1) 1 40 > $P/c5_trace_by_grid.txt
cp $(ls $P/trace_c5/*/*kernel_stats.csv | head -1) $P/c5_kernel_stats.csv
rm -rf $P/trace_c5
head -50 $P/c5_trace_by_grid.txt

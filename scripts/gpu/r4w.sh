#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
cd /tmp; rm -rf /tmp/kt
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python $R/bench.py --cpu-frames 0 --exact-f32-steps 0 > /tmp/kt.out 2> /tmp/kt.err
echo "kernel trace rc $?"; tail -1 /tmp/kt.out > $R/gpurun_out/r4w_bench_config3_under_rocprof.json; cut -c1-200 $R/gpurun_out/r4w_bench_config3_under_rocprof.json
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r4w_config3_kernel_stats.csv && grep -E "finalize|memread_select" $f | cut -c1-200

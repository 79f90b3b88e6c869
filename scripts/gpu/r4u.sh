#!/bin/bash
# round 3 final: rocprofv3 kernel stats of the config-3 bench command, memory-read tests on the final library
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 120 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "memory_read" 2>&1 | tail -2 | cut -c1-200
cd /tmp; rm -rf /tmp/kt
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python $R/bench.py --cpu-frames 0 --exact-f32-steps 0 > /tmp/kt.out 2> /tmp/kt.err
echo "kernel trace rc $?"; tail -1 /tmp/kt.out | cut -c1-300
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r4u_config3_kernel_stats.csv && head -12 $f | cut -c1-170

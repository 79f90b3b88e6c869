#!/bin/bash
# per-shape conv time inside one config-3 session (every launch bracketed by HIP events)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
MIVOS_BENCH_SHAPES=1 timeout 100 python bench.py --steps 137 --warmup 8 --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 1 2> gpurun_out/r4y_conv_shapes.txt > /dev/null
grep -c "^#" gpurun_out/r4y_conv_shapes.txt; grep "^#" gpurun_out/r4y_conv_shapes.txt | head -30 | cut -c1-200

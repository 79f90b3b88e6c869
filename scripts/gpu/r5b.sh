#!/bin/bash
# per-shape conv time with the fusion branch on the main stream (no concurrent kernels)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
MIVOS_FUSE_SIDE_STREAM=0 MIVOS_BENCH_SHAPES=1 timeout 100 python bench.py --steps 137 --warmup 8 --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 1 2> gpurun_out/r5b_conv_shapes_no_side_stream.txt > /dev/null
grep "^#" gpurun_out/r5b_conv_shapes_no_side_stream.txt | head -24 | cut -c1-190

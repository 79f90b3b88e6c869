#!/bin/bash
# last check of the final build: kernel-level GPU tests + a short bench line
set +e
export TMPDIR=/tmp
O=gpurun_out
timeout 200 python -m pytest tests/test_gpu_ops.py -q -x > $O/r3h_ops.log 2>&1; tail -3 $O/r3h_ops.log
timeout 100 python bench.py --steps 137 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 > $O/r3h_bench.json 2> $O/r3h_bench.err; cut -c1-220 $O/r3h_bench.json

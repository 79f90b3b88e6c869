#!/bin/bash
# round 3: where the 256-query select kernel starts to pay (bank depth sweep), memory-read tests, config 5
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "memory_read" 2>&1 | tail -3 | cut -c1-300
timeout 200 python scripts/memread_q256_bench.py --wide 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4s_memread_q256_bench.txt

#!/bin/bash
# hi-first variant of the 256-query select kernel: forced small cases, deep-bank cases, timing
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 40 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "q256hf or deep_bank" -rP 2>&1 | grep -E "hi-first|passed|failed|Error|assert" | cut -c1-260 | tail -8
timeout 45 python scripts/memread_q256_bench.py --hf 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c_memread_hifirst_bench.txt

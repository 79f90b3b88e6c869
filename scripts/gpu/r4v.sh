#!/bin/bash
# finalize launch sized for the plan the select launch used: memory-read tests, end-to-end golden, finalize timing
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py::test_end_to_end_golden -m gpu -q -k "memory_read or golden" 2>&1 | tail -2 | cut -c1-200
timeout 100 python scripts/memread_microbench.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tee gpurun_out/r4v_memread_microbench.txt | awk '{print $1,$2,$3,$4, "finalize", $(NF-4), $(NF-3)}'

#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -m pytest "tests/test_gpu_ops.py::test_memory_read_deep_bank_1080p_vs_chunked_oracle" -m gpu -q -rP 2>&1 | grep -E "^deep bank|^E  |assert|passed|failed" | cut -c1-330 | head -40

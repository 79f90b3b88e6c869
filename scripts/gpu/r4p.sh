#!/bin/bash
# round 3: 480p select kernel: XCD-major chunk order (A/B by environment), threshold as the counting cut's lower bound
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "memory_read" 2>&1 | tail -3 | cut -c1-300
for x in 1 0; do
echo "== MIVOS_MEMREAD_XCD_ORDER=$x"
MIVOS_MEMREAD_XCD_ORDER=$x timeout 150 python scripts/memread_microbench.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee gpurun_out/r4p_memread_microbench_xcd$x.txt | grep 480p
done
for c in "5 7 1620 50" "1 5 1620 20"; do
MIVOS_MEMREAD_DBG=1 timeout 60 python scripts/memread_case.py $c q64 2>&1 | grep "memread_select" | tail -1
done | tee gpurun_out/r4p_memread_cycles.txt

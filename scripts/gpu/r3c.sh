#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "memory_read or split_keys" > $O/r3c_tests.log 2>&1; tail -4 $O/r3c_tests.log
echo "== BR forced, small shapes"
for c in "7 30 54 5 50" "3 9 13 2 50" "23 30 54 1 50" "40 8 10 1 50"; do MIVOS_MEMREAD_BR_MIN=1 timeout 200 python scripts/memread_check.py $c 2>&1 | grep "bad queries"; done
echo "== BR forced, fp32 kernel"; MIVOS_MEMREAD_BR_MIN=1 timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "memory_read" 2>&1 | tail -2
echo "== microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r3c_micro.txt 2>&1

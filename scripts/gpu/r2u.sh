#!/bin/bash
# 32-queries-per-wave fp16 select kernel: correctness on the test shapes, then the three kernels side by side
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "memory_read or split_keys" > $O/r2u_tests.log 2>&1; tail -12 $O/r2u_tests.log
echo "== microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r2u_micro.txt 2>&1; cut -c1-400 $O/r2u_micro.txt
echo "== cycles"; MIVOS_MEMREAD_DBG=1 timeout 300 python scripts/memread_microbench.py > $O/r2u_micro_dbg.txt 2>&1; grep "q128\]" $O/r2u_micro_dbg.txt | awk '{k=$2 $3 $4 $5; if (c[k]++ < 1) print}' | cut -c1-330 | head -12
echo "== skeleton"; MIVOS_ABL=1 timeout 300 python scripts/memread_microbench.py > $O/r2u_micro_abl.txt 2>&1; cut -c1-400 $O/r2u_micro_abl.txt

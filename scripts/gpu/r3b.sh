#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "memory_read or split_keys" > $O/r3b_tests.log 2>&1; tail -4 $O/r3b_tests.log
echo "== microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r3b_micro.txt 2>&1
echo "== cycles"; MIVOS_MEMREAD_DBG=1 timeout 300 python scripts/memread_microbench.py > $O/r3b_micro_dbg.txt 2>&1
echo "== skeleton"; MIVOS_ABL=1 timeout 300 python scripts/memread_microbench.py > $O/r3b_micro_abl.txt 2>&1

#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "memory_read or split_keys" > $O/r2w_tests.log 2>&1; tail -6 $O/r2w_tests.log
echo "== BR small"; MIVOS_MEMREAD_BR_MIN=1 timeout 200 python scripts/memread_check.py 7 30 54 5 50 2>&1 | grep "bad queries"
echo "== BR small 2"; MIVOS_MEMREAD_BR_MIN=1 timeout 200 python scripts/memread_check.py 3 9 13 2 50 2>&1 | grep "bad queries"
echo "== microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r2w_micro.txt 2>&1; cut -c1-105,225-400 $O/r2w_micro.txt
echo "== cycles"; MIVOS_MEMREAD_DBG=1 timeout 300 python scripts/memread_microbench.py > $O/r2w_micro_dbg.txt 2>&1; grep "q128\]" $O/r2w_micro_dbg.txt | awk '{k=$2 $3 $4 $5; if (c[k]++ < 1) print}' | cut -c1-330 | head -12
echo "== skeleton"; MIVOS_ABL=1 timeout 300 python scripts/memread_microbench.py > $O/r2w_micro_abl.txt 2>&1; cut -c1-105,225-400 $O/r2w_micro_abl.txt

#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "memory_read or split_keys" > $O/r2z_tests.log 2>&1; tail -4 $O/r2z_tests.log
echo "== BR small"; MIVOS_MEMREAD_BR_MIN=1 timeout 200 python scripts/memread_check.py 7 30 54 5 50 2>&1 | grep "bad queries"
for A in 8 3 16 32; do
echo "== microbench ahead $A"; MIVOS_MEMREAD_AHEAD=$A timeout 300 python scripts/memread_microbench.py > $O/r2z_micro_a$A.txt 2>&1
echo "== skeleton ahead $A"; MIVOS_MEMREAD_AHEAD=$A MIVOS_ABL=1 timeout 300 python scripts/memread_microbench.py > $O/r2z_micro_abl_a$A.txt 2>&1
done

#!/bin/bash
# f16x3 affinity: correctness of the memory-read tests in both precisions, then the select kernels side by side
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "memory_read or split_keys" > $O/r2s_tests.log 2>&1; tail -12 $O/r2s_tests.log
echo "== microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r2s_micro.txt 2>&1; cat $O/r2s_micro.txt
echo "== cycles"; MIVOS_MEMREAD_DBG=1 timeout 300 python scripts/memread_microbench.py > $O/r2s_micro_dbg.txt 2>&1; grep "f16x3\]" $O/r2s_micro_dbg.txt | awk '!seen[$0]++' | cut -c1-330 | head -12
echo "== skeleton"; MIVOS_ABL=1 timeout 300 python scripts/memread_microbench.py > $O/r2s_micro_abl.txt 2>&1; cat $O/r2s_micro_abl.txt
echo "== branchy everywhere"; MIVOS_MEMREAD_BR_MIN16=1 timeout 300 python scripts/memread_microbench.py > $O/r2s_micro_br.txt 2>&1; cat $O/r2s_micro_br.txt

#!/bin/bash
# state after the f16x3 affinity + 128-query kernel: memread tests, microbench (all three select kernels), bench configs 5 and 3
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "memory_read or split_keys" > $O/r3a_tests.log 2>&1; tail -3 $O/r3a_tests.log
echo "== microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r3a_micro.txt 2>&1; tail -4 $O/r3a_micro.txt | cut -c1-120
echo "== bench config 5"; timeout 900 python bench.py --config 5 > $O/r3a_bench_c5.json 2> $O/r3a_bench_c5.err; cut -c1-200 $O/r3a_bench_c5.json
echo "== bench config 3"; timeout 600 python bench.py > $O/r3a_bench_c3.json 2> $O/r3a_bench_c3.err; cut -c1-200 $O/r3a_bench_c3.json

#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
echo "== memread tests"; timeout 300 python -m pytest tests/test_gpu_ops.py -k memory_read -q 2>&1 | tail -3
echo "== memread microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r2q_memread.txt 2>&1; tail -10 $O/r2q_memread.txt
echo "== 1080p engine test"; timeout 600 python -m pytest tests/test_gpu_engine.py -q -k "1080p" 2>&1 | tail -3
echo "== bench config 5 (1000 frames)"; timeout 900 python bench.py --config 5 --cpu-frames 2 > $O/r2q_bench_c5.json 2> $O/r2q_bench_c5.err; cut -c1-200 $O/r2q_bench_c5.json

#!/bin/bash
# round-2 GPU call L: config 5 full length, config 4 suite (single GPU), 1080p parity tests
set +e
export TMPDIR=/tmp
O=gpurun_out
echo "== 1080p tests"; timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "1080p" -s > $O/r2l_pytest1080.log 2>&1; grep -E "1080p|passed|failed|Error|assert" $O/r2l_pytest1080.log | tail -12
echo "== bench config 5 (1000 frames)"; timeout 900 python bench.py --config 5 --cpu-frames 2 > $O/r2l_bench_c5.json 2> $O/r2l_bench_c5.err; cut -c1-400 $O/r2l_bench_c5.json; tail -3 $O/r2l_bench_c5.err
echo "== bench config 4 (48 clips, 1 GPU)"; timeout 900 python bench.py --config 4 > $O/r2l_bench_c4.json 2> $O/r2l_bench_c4.err; cut -c1-1200 $O/r2l_bench_c4.json; tail -3 $O/r2l_bench_c4.err

#!/bin/bash
# round-2 GPU call A: new memory-read kernel first (microbench + ablation), then the full GPU test-suite and the bench lines
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread microbench" ; timeout 300 python scripts/memread_microbench.py --check > $O/r2a_memread.txt 2>&1 ; tail -12 $O/r2a_memread.txt
echo "== memread ablation (MFMA + staging only)" ; MIVOS_ABL=1 timeout 200 python scripts/memread_microbench.py > $O/r2a_memread_abl1.txt 2>&1 ; tail -10 $O/r2a_memread_abl1.txt
echo "== pytest gpu" ; timeout 900 python -m pytest tests -m gpu -x -q > $O/r2a_pytest.log 2>&1 ; tail -15 $O/r2a_pytest.log
echo "== bench config 3" ; timeout 600 python bench.py > $O/r2a_bench_c3.json 2> $O/r2a_bench_c3.err ; tail -c 1500 $O/r2a_bench_c3.json; tail -5 $O/r2a_bench_c3.err
echo "== bench config 3 driver flags" ; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 > $O/r2a_bench_c3_driver.json 2> $O/r2a_bench_c3_driver.err ; cut -c1-400 $O/r2a_bench_c3_driver.json
echo "== bench config 2" ; timeout 300 python bench.py --config 2 > $O/r2a_bench_c2.json 2> $O/r2a_bench_c2.err ; cut -c1-600 $O/r2a_bench_c2.json; tail -3 $O/r2a_bench_c2.err
echo "== bench config 5 (260 frames)" ; timeout 600 python bench.py --config 5 --frames 260 --cpu-frames 0 > $O/r2a_bench_c5_260.json 2> $O/r2a_bench_c5_260.err ; cut -c1-1500 $O/r2a_bench_c5_260.json; tail -5 $O/r2a_bench_c5_260.err

#!/bin/bash
# round-2 GPU call B: memory read v4 (prefetch-pipelined vs plain), then the full GPU test-suite and the bench lines
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests (PF=1)"; timeout 300 python -m pytest tests/test_gpu_ops.py -k memory_read -x -q > $O/r2b_memtest_pf1.log 2>&1; rc1=$?; tail -5 $O/r2b_memtest_pf1.log
echo "== memread tests (PF=0)"; MIVOS_MEMREAD_PF=0 timeout 300 python -m pytest tests/test_gpu_ops.py -k memory_read -x -q > $O/r2b_memtest_pf0.log 2>&1; rc0=$?; tail -5 $O/r2b_memtest_pf0.log
echo "== memread microbench PF=1"; timeout 300 python scripts/memread_microbench.py --check > $O/r2b_memread_pf1.txt 2>&1; tail -11 $O/r2b_memread_pf1.txt
echo "== memread microbench PF=0"; MIVOS_MEMREAD_PF=0 timeout 300 python scripts/memread_microbench.py > $O/r2b_memread_pf0.txt 2>&1; tail -10 $O/r2b_memread_pf0.txt
echo "== memread ablation PF=1 (MFMA + staging only)"; MIVOS_ABL=1 timeout 200 python scripts/memread_microbench.py > $O/r2b_memread_abl1.txt 2>&1; tail -10 $O/r2b_memread_abl1.txt
if [ $rc1 -ne 0 ]; then
  if [ $rc0 -ne 0 ]; then echo "both memread variants fail: stopping"; exit 0; fi
  export MIVOS_MEMREAD_PF=0; echo "PF=1 fails, continuing with PF=0"
fi
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q > $O/r2b_pytest.log 2>&1; tail -25 $O/r2b_pytest.log
echo "== bench config 3"; timeout 600 python bench.py > $O/r2b_bench_c3.json 2> $O/r2b_bench_c3.err; tail -c 1800 $O/r2b_bench_c3.json; tail -5 $O/r2b_bench_c3.err
echo "== bench config 3 driver flags"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 > $O/r2b_bench_c3_driver.json 2> $O/r2b_bench_c3_driver.err; cut -c1-400 $O/r2b_bench_c3_driver.json
echo "== bench config 2"; timeout 300 python bench.py --config 2 > $O/r2b_bench_c2.json 2> $O/r2b_bench_c2.err; cut -c1-600 $O/r2b_bench_c2.json; tail -3 $O/r2b_bench_c2.err
echo "== bench config 5 (260 frames)"; timeout 600 python bench.py --config 5 --frames 260 --cpu-frames 0 > $O/r2b_bench_c5_260.json 2> $O/r2b_bench_c5_260.err; cut -c1-1500 $O/r2b_bench_c5_260.json; tail -5 $O/r2b_bench_c5_260.err

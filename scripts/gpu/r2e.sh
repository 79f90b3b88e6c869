#!/bin/bash
# round-2 GPU call E: memory read v4 with branch-free sliced selection; cycles per tile; then tests + benches
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 300 python -m pytest tests/test_gpu_ops.py -k memory_read -q > $O/r2e_memtest.log 2>&1; rc1=$?; tail -6 $O/r2e_memtest.log
echo "== memread microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r2e_memread.txt 2>&1; tail -11 $O/r2e_memread.txt
echo "== memread ablation (MFMA + staging only)"; MIVOS_ABL=1 timeout 200 python scripts/memread_microbench.py > $O/r2e_memread_abl1.txt 2>&1; tail -10 $O/r2e_memread_abl1.txt
echo "== cycles per tile"; MIVOS_MEMREAD_DBG=1 timeout 200 python scripts/memread_microbench.py 2>&1 | grep -E "memread_select\]" | awk 'NR%6==0' > $O/r2e_memread_cycles.txt; cat $O/r2e_memread_cycles.txt
echo "== cycles per tile, ablation"; MIVOS_ABL=1 MIVOS_MEMREAD_DBG=1 timeout 200 python scripts/memread_microbench.py 2>&1 | grep -E "memread_select\]" | awk 'NR%6==0' > $O/r2e_memread_cycles_abl1.txt; cat $O/r2e_memread_cycles_abl1.txt
if [ $rc1 -ne 0 ]; then echo "memread tests fail: stopping"; exit 0; fi
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q > $O/r2e_pytest.log 2>&1; tail -25 $O/r2e_pytest.log
echo "== bench config 3"; timeout 600 python bench.py > $O/r2e_bench_c3.json 2> $O/r2e_bench_c3.err; tail -c 1800 $O/r2e_bench_c3.json; tail -5 $O/r2e_bench_c3.err
echo "== bench config 3 driver flags"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 > $O/r2e_bench_c3_driver.json 2> $O/r2e_bench_c3_driver.err; cut -c1-400 $O/r2e_bench_c3_driver.json
echo "== bench config 2"; timeout 300 python bench.py --config 2 > $O/r2e_bench_c2.json 2> $O/r2e_bench_c2.err; cut -c1-600 $O/r2e_bench_c2.json; tail -3 $O/r2e_bench_c2.err
echo "== bench config 5 (260 frames)"; timeout 600 python bench.py --config 5 --frames 260 --cpu-frames 0 > $O/r2e_bench_c5_260.json 2> $O/r2e_bench_c5_260.err; cut -c1-1500 $O/r2e_bench_c5_260.json; tail -5 $O/r2e_bench_c5_260.err

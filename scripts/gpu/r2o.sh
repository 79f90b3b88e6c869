#!/bin/bash
# round-2 GPU call O: multi-rank plumbing of bench.py on one GPU (gloo), full gpu suite, final bench lines
set +e
export TMPDIR=/tmp
O=gpurun_out
echo "== bench --gpus 2 self-spawn (gloo, both ranks on cuda:0)"; MIVOS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 40 --warmup 10 > $O/r2o_bench_2rank_gloo.json 2> $O/r2o_bench_2rank_gloo.err; cut -c1-300 $O/r2o_bench_2rank_gloo.json; tail -3 $O/r2o_bench_2rank_gloo.err
echo "== bench --gpus 2 --config 4 (gloo)"; MIVOS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config 4 --clips 8 > $O/r2o_bench_c4_2rank_gloo.json 2> $O/r2o_bench_c4_2rank_gloo.err; cut -c1-1500 $O/r2o_bench_c4_2rank_gloo.json; tail -3 $O/r2o_bench_c4_2rank_gloo.err
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/r2o_pytest.log 2>&1; tail -14 $O/r2o_pytest.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench config 3 (default)"; timeout 600 python bench.py > $O/r2o_bench_c3.json 2> $O/r2o_bench_c3.err; cut -c1-200 $O/r2o_bench_c3.json
echo "== bench config 3 driver flags"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2o_bench_c3_driver.json 2> $O/r2o_bench_c3_driver.err; cut -c1-200 $O/r2o_bench_c3_driver.json
echo "== bench config 2"; timeout 300 python bench.py --config 2 > $O/r2o_bench_c2.json 2> $O/r2o_bench_c2.err; cut -c1-200 $O/r2o_bench_c2.json

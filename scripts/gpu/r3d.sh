#!/bin/bash
# round-2 final state: full GPU suite, bench configs 3 / 5 / 2
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/r3d_pytest.log 2>&1; tail -10 $O/r3d_pytest.log
echo "== bench config 3"; timeout 600 python bench.py > $O/r3d_bench_c3.json 2> $O/r3d_bench_c3.err; cut -c1-200 $O/r3d_bench_c3.json
echo "== bench config 5"; timeout 900 python bench.py --config 5 > $O/r3d_bench_c5.json 2> $O/r3d_bench_c5.err; cut -c1-200 $O/r3d_bench_c5.json
echo "== bench config 2"; timeout 600 python bench.py --config 2 > $O/r3d_bench_c2.json 2> $O/r3d_bench_c2.err; cut -c1-200 $O/r3d_bench_c2.json

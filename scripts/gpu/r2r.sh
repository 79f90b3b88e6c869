#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/r2r_pytest.log 2>&1; tail -10 $O/r2r_pytest.log
echo "== bench config 5 (1000 frames)"; timeout 900 python bench.py --config 5 > $O/r2r_bench_c5.json 2> $O/r2r_bench_c5.err; cut -c1-200 $O/r2r_bench_c5.json

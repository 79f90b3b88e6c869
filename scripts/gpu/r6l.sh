#!/bin/bash
# round 5, call 12: FusionGenerator's two passes on two streams: tests, then the generator suite with / without
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest -q tests/test_gpu_engine.py -m gpu -k "generator" > gpurun_out/r6l_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r6l_pytest.log | cut -c1-250
for e in "MIVOS_CONCURRENT_PASSES=1" "MIVOS_CONCURRENT_PASSES=0" "MIVOS_CONCURRENT_PASSES=1" "MIVOS_CONCURRENT_PASSES=0"; do
  env $e timeout 400 python bench.py --config 4 --generator --clips 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e generator 8 clips:', d['value'], d['unit'])" | tee -a gpurun_out/r6l_generator_ab.txt
done

#!/bin/bash
# round 3, call 8: adam fix, training bench (1 rank; 2 ranks over gloo sharing the GPU = plumbing of the data-parallel step)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q > gpurun_out/r4h_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r4h_pytest.log
timeout 300 python bench.py --config train 2> gpurun_out/r4h_train.err | tee gpurun_out/r4h_bench_train.json | cut -c1-900
tail -3 gpurun_out/r4h_train.err
MIVOS_DIST_BACKEND=gloo timeout 300 python bench.py --config train --gpus 2 --steps 4 2> gpurun_out/r4h_train2.err | tee gpurun_out/r4h_bench_train_2ranks_gloo_one_gpu.json | cut -c1-1200
tail -3 gpurun_out/r4h_train2.err
cd /tmp; rm -rf /tmp/ks
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python $OLDPWD/bench.py --config train --steps 6 > /dev/null 2> /tmp/ks.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $OLDPWD/gpurun_out/r4h_train_kernel_stats.csv; head -14 "$f" | cut -c1-150

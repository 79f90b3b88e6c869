#!/bin/bash
# round 3, call 4: training-step tests, conv1-from-planes + pipelined resblock, benches, rocprof stats
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rP > gpurun_out/r4d_pytest_train.log 2>&1
echo "pytest train rc $?"; tail -4 gpurun_out/r4d_pytest_train.log; grep -E "^it |Error|error|assert" gpurun_out/r4d_pytest_train.log | head -20
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -q -x -k "fusion_resblock or fusion_head or fusion_conv1 or fusion_net_forward or fusion_net_golden or mem_profiles or end_to_end" > gpurun_out/r4d_pytest.log 2>&1
echo "pytest rc $?"; tail -2 gpurun_out/r4d_pytest.log
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session"
sumline() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], r['kernel'], r['frac'], 'aff_us', r['affinity']['avg_launch_us'], {k[:24]:(v['launches'],v['avg_us']) for k,v in d['conv_kernels'].items() if 'fusion' in k or 'direct' in k})"; }
timeout 200 $B 2> gpurun_out/r4d_a.err | tee gpurun_out/r4d_bench_side_stream.json | sumline side_stream
MIVOS_FUSE_SIDE_STREAM=0 timeout 200 $B 2>/dev/null | tee gpurun_out/r4d_bench_no_side_stream.json | sumline no_side_stream
cd /tmp
rm -rf /tmp/ks
MIVOS_FUSE_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python $R/bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 274 > $R/gpurun_out/r4d_stats_bench.json 2> /tmp/ks.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r4d_config3_kernel_stats.csv
grep -E "fusion|interleave|direct" $R/gpurun_out/r4d_config3_kernel_stats.csv | cut -c1-140
cd $R
timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driverflags', d['value'], d['ms_per_step'], d['full_session'])"

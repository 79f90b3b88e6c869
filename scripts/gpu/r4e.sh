#!/bin/bash
# round 3, call 5: training tests (fixed), full-softmax read, conv1-from-planes v2, async query prefetch A/B
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rP > gpurun_out/r4e_pytest_train.log 2>&1
echo "pytest train rc $?"; tail -3 gpurun_out/r4e_pytest_train.log; grep -E "^it " gpurun_out/r4e_pytest_train.log | head
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -q -rP -k "full_softmax or no_topk or fusion_conv1 or fusion_net_forward or fusion_net_golden or mem_profiles or end_to_end or callbacks or reinteraction or 480p_propagation" > gpurun_out/r4e_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r4e_pytest.log; grep -E "^full softmax|^no_topk|worst per-frame" gpurun_out/r4e_pytest.log | cut -c1-260
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session"
sumline() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], r['kernel'], r['frac'], 'aff_us', r['affinity']['avg_launch_us'], {k[:24]:(v['launches'],v['avg_us']) for k,v in d['conv_kernels'].items() if 'fusion' in k or 'direct' in k})"; }
timeout 200 $B 2> gpurun_out/r4e_a.err | tee gpurun_out/r4e_bench_prefetch.json | sumline prefetch_side
MIVOS_QUERY_PREFETCH=0 timeout 200 $B 2>/dev/null | tee gpurun_out/r4e_bench_no_prefetch.json | sumline no_prefetch_side
MIVOS_QUERY_PREFETCH=0 MIVOS_FUSE_SIDE_STREAM=0 timeout 200 $B 2>/dev/null | sumline no_prefetch_no_side
MIVOS_QUERY_BATCH=16 timeout 200 $B 2>/dev/null | sumline prefetch_side_qb16
timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driverflags', d['value'], d['ms_per_step'], d['full_session'])"
timeout 200 python bench.py --config 2 --cpu-frames 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2', d['value'], d['ms_per_step'], d['full_session'])"
cd /tmp; rm -rf /tmp/ks
MIVOS_QUERY_PREFETCH=0 MIVOS_FUSE_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python $R/bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 274 > $R/gpurun_out/r4e_stats_bench.json 2> /tmp/ks.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r4e_config3_kernel_stats.csv
grep -E "fusion|interleave" $R/gpurun_out/r4e_config3_kernel_stats.csv | cut -c1-140

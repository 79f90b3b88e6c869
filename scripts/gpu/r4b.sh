#!/bin/bash
# round 3, call 2: fused FusionNet kernels, in-kernel split-K fix-up, side-stream fusion: targeted tests + A/B bench lines
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -rP -k "fusion or conv2d or sh32 or lds_dma" > gpurun_out/r4b_pytest_ops.log 2>&1
echo "pytest ops rc $?"; tail -3 gpurun_out/r4b_pytest_ops.log; grep -E "^fusion_resblock|^FusionNet" gpurun_out/r4b_pytest_ops.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -rP -k "fusion or golden or headline or 480p_propagation or reinteraction or mem_profiles" > gpurun_out/r4b_pytest_engine.log 2>&1
echo "pytest engine rc $?"; tail -3 gpurun_out/r4b_pytest_engine.log; grep -E "^FusionNet|worst per-frame" gpurun_out/r4b_pytest_engine.log | cut -c1-260
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session"
sumline() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], r['kernel'], r['frac'], 'aff_us', r['affinity']['avg_launch_us'], {k:(v['launches'],v['avg_us']) for k,v in d['conv_kernels'].items() if 'fusion' in k or 'direct' in k})"; }
timeout 200 $B 2> gpurun_out/r4b_a.err | tee gpurun_out/r4b_bench_all_new.json | sumline all_new
MIVOS_FUSE_SIDE_STREAM=0 timeout 200 $B 2>/dev/null | tee gpurun_out/r4b_bench_no_side_stream.json | sumline no_side_stream
MIVOS_FUSE_SIDE_STREAM=0 MIVOS_PP_SPLITK_TWO_PASS=1 timeout 200 $B 2>/dev/null | tee gpurun_out/r4b_bench_no_side_two_pass.json | sumline no_side_two_pass
MIVOS_FUSE_SIDE_STREAM=0 MIVOS_PP_SPLITK_TWO_PASS=1 MIVOS_FUSION_ONE_CALL=0 timeout 200 $B 2>/dev/null | tee gpurun_out/r4b_bench_old_paths.json | sumline old_paths
timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driverflags', d['value'], d['ms_per_step'], d['full_session'])"
MIVOS_QUERY_BATCH=16 timeout 200 $B 2>/dev/null | sumline qb16

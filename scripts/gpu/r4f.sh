#!/bin/bash
# round 3, call 6: stem-from-planes kernel, FusionGenerator, generator bench; A/B
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_train.py -m gpu -q -rP -k "stem or fusion_generator or query_encoder or 480p_single_step or end_to_end or segment_with_query or adam or attention_read" > gpurun_out/r4f_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r4f_pytest.log; grep -E "^stem |fusion_generator|^E  " gpurun_out/r4f_pytest.log | cut -c1-260 | head -20
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session"
sumline() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], r['kernel'], r['frac'], 'aff_us', r['affinity']['avg_launch_us'])"; }
timeout 200 $B 2> gpurun_out/r4f_a.err | tee gpurun_out/r4f_bench_stem.json | sumline stem_planes
MIVOS_STEM_PLANES=0 timeout 200 $B 2>/dev/null | tee gpurun_out/r4f_bench_no_stem.json | sumline old_stem
timeout 200 $B 2>/dev/null | sumline stem_planes_again
timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driverflags', d['value'], d['ms_per_step'], d['full_session'])"
timeout 300 python bench.py --config 4 --generator --clips 8 2> gpurun_out/r4f_gen.err | tee gpurun_out/r4f_bench_generator.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('generator', d['value'], d['ms_per_step'], d['steps'], d['config']['reference_frames'])"
timeout 300 python bench.py --config 4 --clips 48 2>/dev/null | tee gpurun_out/r4f_bench_config4_48.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['value'], d['ms_per_step'], d['steps'])"
cd /tmp; rm -rf /tmp/ks
MIVOS_FUSE_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python $R/bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 274 > $R/gpurun_out/r4f_stats_bench.json 2> /tmp/ks.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r4f_config3_kernel_stats.csv
grep -E "stem|maxpool|interleave|conv_f16x3_kernel" $R/gpurun_out/r4f_config3_kernel_stats.csv | cut -c1-150

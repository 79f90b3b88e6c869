#!/bin/bash
# A/B: M = 8100 long-K layers on split-K 128x256 tiles (MIVOS_PP_WIDE_NK) vs 128x128 tiles
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 96 64 0 96; do
MIVOS_PP_WIDE_NK=$v timeout 60 python bench.py --steps 274 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 --no-full-session 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('wide_nk $v', d['value'], d['ms_per_step'], {k:(v['avg_us'],v['time_share']) for k,v in d['conv_kernels'].items() if 'pp_kernel' in k})"
done 2>&1 | tee gpurun_out/r4z_wide_split_ab.txt

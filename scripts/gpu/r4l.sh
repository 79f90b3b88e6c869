#!/bin/bash
# round 3: the 256-queries-per-workgroup select kernel: forced on the small oracle cases, the deep-bank cases, timing vs the 128-query kernel, config 5
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "q256" 2>&1 | tail -3 | cut -c1-300
timeout 240 python -m pytest tests/test_gpu_ops.py -m gpu -q -rP -k "deep_bank" 2>&1 | grep -E "^deep bank|passed|failed|Error" | cut -c1-330
timeout 120 python scripts/memread_q256_bench.py 2>&1 | tail -4 | tee gpurun_out/r4l_memread_q256_bench.txt
MIVOS_MEMREAD_Q256_MIN=400000 timeout 200 python bench.py --config 5 --cpu-frames 0 2>/dev/null | tee gpurun_out/r4l_bench_config5_q256.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('config5 q256', d['value'], d['ms_per_step'], r['affinity']['avg_launch_us'], r['affinity']['frac_of_f32_mfma_peak'])"

#!/bin/bash
# round 3: 256-query select kernel without the compiler's vmcnt(0) per tile, four scores per threshold test, one accumulator chain
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "q256 or deep_bank" 2>&1 | tail -3 | cut -c1-300
timeout 120 python scripts/memread_q256_bench.py 2>&1 | tail -4 | tee gpurun_out/r4q_memread_q256_bench.txt
timeout 200 python bench.py --config 5 --cpu-frames 0 2>/dev/null | tail -1 > gpurun_out/r4q_bench_config5.json
python -c "
import json
d=json.loads(open('gpurun_out/r4q_bench_config5.json').read()); r=d['roofline']['affinity']; print('config5', d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac_of_f32_mfma_peak'])"

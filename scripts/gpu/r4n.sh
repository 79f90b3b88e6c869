#!/bin/bash
# round 3: 480p select kernel with the counting cut (count_kth) and the pointer fill level: oracle cases, timing, configs 3 / 2
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "memory_read" 2>&1 | tail -4 | cut -c1-300
timeout 150 python scripts/memread_microbench.py --check 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4n_memread_microbench.txt | cut -c1-330
for c in 3 2; do
timeout 200 python bench.py --config $c --cpu-frames 0 --exact-f32-steps 0 2>/dev/null | tail -1 > gpurun_out/r4n_bench_config$c.json
python -c "
import json
d=json.loads(open('gpurun_out/r4n_bench_config$c.json').read()); r=d['roofline']['affinity']; print('config$c', d['value'], d['ms_per_step'], d.get('full_session'), r['avg_launch_us'], r['frac_of_f32_mfma_peak'])" | cut -c1-600
done

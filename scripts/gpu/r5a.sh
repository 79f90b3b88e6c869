#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 80 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 2>gpurun_out/r5a.err | tail -1 > gpurun_out/r5a_bench_driverflags.json
python -c "
import json
d=json.loads(open('gpurun_out/r5a_bench_driverflags.json').read()); r=d['roofline']; print(d['value'], r['frac'], r['avg_launch_us'], r['isolated'], r['mfma_util_pmc'], d['full_session']['value'])"
tail -3 gpurun_out/r5a.err | cut -c1-300

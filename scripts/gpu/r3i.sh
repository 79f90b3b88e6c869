#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
for v in 1 0 1 0; do
MIVOS_FUSION_ONE_CALL=$v timeout 100 python bench.py --steps 411 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one_call=$v', d['value'], d['ms_per_step'])"
done

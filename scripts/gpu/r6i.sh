#!/bin/bash
# round 5, call 9: three sessions of five objects in flight again, now that the kernels know how many streams share the chip (chip_share)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
for L in 2 3 2 3 4; do
  timeout 200 $B --lanes $L --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3 lanes', d['config']['clips_in_flight_per_gpu'], d['value'], (d.get('one_clip_in_flight') or {}).get('value'))" | tee -a gpurun_out/r6i_lanes_config3.txt
done

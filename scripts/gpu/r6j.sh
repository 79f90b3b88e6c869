#!/bin/bash
# round 5, call 10: two identical sessions in lockstep vs staggered by a few frames (query batches of 10 frames)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
for S in 0 5 0 5 3 7; do
  MIVOS_LANE_STAGGER=$S timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3 lanes 2 stagger $S:', d['value'], (d.get('one_clip_in_flight') or {}).get('value'))" | tee -a gpurun_out/r6j_stagger.txt
done

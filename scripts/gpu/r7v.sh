#!/bin/bash
# round 6: clips in flight for the suite (config 4, first 96 clips) and for config 3 - 2 / 3 / 4 lanes on the final tree, same box
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for i in 1 2; do for L in 2 3 4; do
  timeout 400 python bench.py --config 4 --clips 96 --lanes $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4 96 clips lanes $L', d['value'], d['config']['suite_checksum'])" | tee -a gpurun_out/r7v_lanes.txt
done; done
for L in 2 3; do
  timeout 300 python bench.py --config 3 --steps 274 --warmup 137 --lanes $L --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-sustained --no-full-session 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3 lanes $L: one clip', d['value'], 'several', d['several_clips_in_flight']['value'])" | tee -a gpurun_out/r7v_lanes.txt
done

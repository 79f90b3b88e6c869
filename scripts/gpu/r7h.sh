#!/bin/bash
# round 6, call 8: HIP hardware queues (GPU_MAX_HW_QUEUES: the lanes use 5-7 streams on the default 4 queues) and the host's cyclic GC (2-3 ms launch stalls in the timeline) - same-box A/B
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
new() {  # name, config args..., then env after --
  name=$1; shift; args=""; while [ "$1" != "--" ]; do args="$args $1"; shift; done; shift
  env "$@" timeout 300 python bench.py $args --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-sustained --no-full-session 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name: one clip', d['value'], 'several', (d.get('several_clips_in_flight') or {}).get('value'))" >> gpurun_out/r7h_ab.txt
}
C3="--config 3 --steps 274 --warmup 137 --lanes 2"
C2="--config 2 --steps 276 --warmup 69 --lanes 3"
for i in 1 2; do
  new "config3 base" $C3 -- X=1
  new "config3 GPU_MAX_HW_QUEUES=8" $C3 -- GPU_MAX_HW_QUEUES=8
  new "config3 GPU_MAX_HW_QUEUES=2" $C3 -- GPU_MAX_HW_QUEUES=2
  new "config2 base" $C2 -- X=1
  new "config2 GPU_MAX_HW_QUEUES=8" $C2 -- GPU_MAX_HW_QUEUES=8
done
for i in 1 2; do
  timeout 400 python bench.py --config 4 --clips 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4 48 clips lanes 3 base', d['value'])" >> gpurun_out/r7h_ab.txt
  GPU_MAX_HW_QUEUES=8 timeout 400 python bench.py --config 4 --clips 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4 48 clips lanes 3 GPU_MAX_HW_QUEUES=8', d['value'])" >> gpurun_out/r7h_ab.txt
done
for i in 1 2; do
  timeout 300 python -c "
import gc, runpy, sys
gc.disable()
sys.argv = ['bench.py'] + '--config 3 --steps 274 --warmup 137 --lanes 1 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-sustained --no-full-session'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3 one clip, cyclic GC disabled', d['value'])" >> gpurun_out/r7h_ab.txt
  new "config3 one clip base" --config 3 --steps 274 --warmup 137 --lanes 1 -- X=1
done
cat gpurun_out/r7h_ab.txt

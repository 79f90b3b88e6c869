#!/bin/bash
# round 5, call 13: with two sessions in flight, do the big GEMMs do better on 128x128 tiles (two workgroups per CU: lanes interleave on a CU) than on 128x256 (one per CU)?
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
for e in "X=1" "MIVOS_PP_TILE=20" "X=1" "MIVOS_PP_TILE=20"; do
  env $e timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'], (d.get('one_clip_in_flight') or {}).get('value'))" | tee -a gpurun_out/r6m_tile20_lanes2.txt
done

#!/bin/bash
# round 6: select-kernel knobs at 480p with one clip in flight (wave-uniform append skip from fewer positions; persistent workgroup count) - same-box A/B
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --config 3 --lanes 1 --steps 274 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 --profile-every 7 --no-sustained --no-full-session 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['roofline']['affinity']; print('$name', d['value'], 'select us', a['avg_launch_us'], 'finalize us', a['finalize']['avg_launch_us'])" | tee -a gpurun_out/r7r_select_knobs.txt; }
for i in 1 2; do
  run base X=1
  run br_min_8192 MIVOS_MEMREAD_BR_MIN=8192
  run br_min_16384 MIVOS_MEMREAD_BR_MIN=16384
  run wgs_192 MIVOS_MEMREAD_WGS=192
  run wgs_224 MIVOS_MEMREAD_WGS=224
done

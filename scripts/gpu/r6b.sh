#!/bin/bash
# round 5, call 2: (1) where the default precision's engine-vs-fp64 distance comes from (one arithmetic switched at a time, 24 frames of the 70-frame clip,
# oracles from gpurun_in/), (2) config 3 with 1 / 2 / 3 sessions in flight, (3) in-situ A/B of the convolution tile rules the per-shape sweep suggested
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
if [ -f gpurun_in/d32/done ]; then
  D="--frames 24 --clip-frames 70 --ref32 gpurun_in/d32 --ref64 gpurun_in/d64 --wait 5"
  rm -f gpurun_out/r6b_diag.txt
  run() { name=$1; shift; timeout 200 python scripts/long_session_parity.py engine $D "$@" --json gpurun_out/r6b_diag_$name.json 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$name', 'interact', d['interact'], 'e/r max', d.get('median_ratio_of_maxima'), 'q999', d.get('median_ratio_of_q999'), 'worst', d.get('worst_ratio_of_maxima'), 'min IoU', d['min_iou'], 'ref32-vs-64 min IoU', d.get('min_iou_ref32_vs_fp64'))" | tee -a gpurun_out/r6b_diag.txt; }
  run default
  run affinity_f32 --affinity f32
  run no_act_path --no-act-path
  run no_stem --no-stem-kernel
  run no_proj --no-cout1-projection
  run exact_f32 --precision f32
  run exact_conv_f16x3_affinity --precision f32 --affinity f16x3
fi
el diag
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
ab() {   # name, env...
  name=$1; shift
  a=$(env "$@" timeout 200 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], (d.get('one_clip_in_flight') or {}).get('value'))")
  b=$(env "$@" timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], (d.get('one_clip_in_flight') or {}).get('value'))")
  echo "$name | driver window: $a | 2 sessions: $b | t=$(( $(date +%s) - t0 ))" | tee -a gpurun_out/r6b_ab.txt
}
rm -f gpurun_out/r6b_ab.txt
ab base0 X=1
ab lanes2 MIVOS_BENCH_LANES=2
ab lanes3 MIVOS_BENCH_LANES=3
ab small128 MIVOS_PP_SMALL_WGS=128
ab longnk200 MIVOS_PP_SPLIT_LONG_NK=200
ab merge MIVOS_PP_MERGE=1
ab base1 X=1
ab small128_longnk MIVOS_PP_SMALL_WGS=128 MIVOS_PP_SPLIT_LONG_NK=200
ab lanes2_small128_longnk MIVOS_BENCH_LANES=2 MIVOS_PP_SMALL_WGS=128 MIVOS_PP_SPLIT_LONG_NK=200
el ab
echo "total $(( $(date +%s) - t0 )) s"

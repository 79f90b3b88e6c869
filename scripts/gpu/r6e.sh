#!/bin/bash
# round 5, call 5: the long session on the admitted fixture (mask gain 0.2, logit gain 0.6; reference fp32 vs fp64 >= 0.9995 at all 137 steps) in f16x3 and exact fp32,
# select-kernel grid A/B with two sessions in flight, one default bench line with the final bench code
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
if [ -f gpurun_in/M32/done ]; then
  for prec in f16x3 f32; do
    timeout 400 python scripts/long_session_parity.py engine --precision $prec --ref32 gpurun_in/M32 $( [ -f gpurun_in/M64/done ] && echo "--ref64 gpurun_in/M64" ) --wait 5 \
      --json gpurun_out/r6e_long_session_parity_$prec.json 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$prec', {k: d.get(k) for k in ('interact', 'min_iou', 'mean_iou', 'frames_below_0999', 'total_mismatch_px', 'max_dprob', 'min_iou_ref32_vs_fp64', 'min_iou_engine_vs_fp64', 'median_ratio_of_maxima', 'median_ratio_of_q999', 'worst_ratio_of_maxima', 'frames_ref32_vs_fp64_below_09995')})"
  done
fi
el long
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
ab() {
  name=$1; shift
  b=$(env "$@" timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], (d.get('one_clip_in_flight') or {}).get('value'))")
  echo "$name | 2 sessions: $b | t=$(( $(date +%s) - t0 ))" | tee -a gpurun_out/r6e_ab.txt
}
rm -f gpurun_out/r6e_ab.txt
ab warm X=1
ab default X=1
ab memread_wgs128 MIVOS_MEMREAD_WGS=128
ab memread_wgs192 MIVOS_MEMREAD_WGS=192
ab default_b X=1
ab memread_wgs160 MIVOS_MEMREAD_WGS=160
el ab
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6e_bench_driverflags.json 2> gpurun_out/r6e_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r6e_bench_driverflags.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], 'one lane', (d.get('one_clip_in_flight') or {}).get('value'), 'full', {k: d['full_session'].get(k) for k in ('value', 'sessions_in_flight')}, (d['full_session'].get('one_clip_in_flight') or {}).get('value'))
print('roof', d['roofline'].get('frac'), d['roofline'].get('timed_region', {}).get('frac'), 'aff', d['roofline'].get('affinity', {}).get('frac'), d['roofline'].get('traffic'))
print('parity', {k: v for k, v in d.get('parity', {}).items() if k != 'fp64'}, {k: v for k, v in d.get('parity', {}).get('fp64', {}).items() if not k.startswith('per_frame')})"
tail -2 gpurun_out/r6e_bench.err
el bench
echo "total $(( $(date +%s) - t0 )) s"

#!/bin/bash
# round 6, call 2: the new bench line on the driver's flags; same-box A/B of the cheap levers with ONE clip in flight (no split-K, fusion off its side
# stream, larger query batches) and of the split-K rules with two clips in flight
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r7b_bench_driverflags.json 2> gpurun_out/r7b_bench_driverflags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7b_bench_driverflags.json').read().strip().splitlines()[-1])
print('driver flags:', d['value'], d['ms_per_step'], 'clocks', d.get('clocks'), '\n several', (d.get('several_clips_in_flight') or {}).get('value'), '\n full', d['full_session']['value'], (d['full_session'].get('several_clips_in_flight') or {}).get('value'),
      '\n sustained', {k: v for k, v in d['sustained'].items() if k != 'note'}, '\n roof', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['timed_region'], '\n parity', d['parity']['min_iou_engine_vs_ref_fp32'], d['parity']['fp64']['gate_passed'], 'cpu', d['cpu_baseline']['value'])
PY
ARGS="--config 3 --steps 274 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-full-session --no-sustained"
ab() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS $LANES 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('several_clips_in_flight') or {}; print('$name', d['value'], d['ms_per_step'], 'several', s.get('value'))" >> gpurun_out/r7b_ab.txt
}
for i in 1 2; do
  LANES="--lanes 1"
  ab base X=1
  ab nosplit MIVOS_PP_SPLIT_THR=0
  ab fuse_main_stream MIVOS_FUSE_SIDE_STREAM=0
  ab qbatch23 MIVOS_QUERY_BATCH=23
  ab qbatch35 MIVOS_QUERY_BATCH=35
  ab nosplit_qbatch23 MIVOS_PP_SPLIT_THR=0 MIVOS_QUERY_BATCH=23
  ab nosplit_fusemain_qb23 MIVOS_PP_SPLIT_THR=0 MIVOS_QUERY_BATCH=23 MIVOS_FUSE_SIDE_STREAM=0
done
LANES="--lanes 2"
for i in 1 2; do
  ab lanes2_base X=1
  ab lanes2_nosplit MIVOS_PP_SPLIT_THR=0
  ab lanes2_sharecap0 MIVOS_PP_SHARE_CAP=0
done
cat gpurun_out/r7b_ab.txt

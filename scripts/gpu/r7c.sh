#!/bin/bash
# round 6, call 3: GPU suite on the refactored host code (per-owner range status, per-step device context, shared side streams); in-situ per-shape diff of
# "no split-K" against the default rules; shape-only split-K candidates (MIVOS_PP_SPLIT_MIN_NK with the chip-share hint ignored) with one and two clips in flight
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r7c_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r7c_pytest.log | cut -c1-300
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r7c_bench_driverflags.json 2> gpurun_out/r7c_bench_driverflags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7c_bench_driverflags.json').read().strip().splitlines()[-1])
print('driver flags:', d['value'], d['ms_per_step'], 'several', (d.get('several_clips_in_flight') or {}).get('value'), 'full', d['full_session']['value'], (d['full_session'].get('several_clips_in_flight') or {}).get('value'),
      'sustained', d['sustained']['value'], d['sustained']['several_clips_in_flight']['value'], 'hbm', d['hbm_peak_allocated_gb'], 'roof', d['roofline']['frac'], d['roofline']['timed_region']['frac'])
PY
cd /tmp
ARGS="--config 3 --lanes 1 --steps 274 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-full-session --no-sustained"
for tag in default nosplit; do
  rm -rf /tmp/kt_$tag /tmp/conv_$tag.log
  if [ $tag = nosplit ]; then export MIVOS_PP_SPLIT_THR=0; else unset MIVOS_PP_SPLIT_THR; fi
  MIVOS_CONV_LOG=/tmp/conv_$tag.log timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$tag --output-format csv -- python $R/bench.py $ARGS > /dev/null 2> /tmp/kt_$tag.err
  python $R/scripts/insitu_shape_table.py /tmp/kt_$tag /tmp/conv_$tag.log --label $tag --json $R/gpurun_out/r7c_insitu_$tag.json > $R/gpurun_out/r7c_insitu_$tag.txt 2>&1
done
unset MIVOS_PP_SPLIT_THR
python $R/scripts/insitu_shape_table.py --diff $R/gpurun_out/r7c_insitu_default.json $R/gpurun_out/r7c_insitu_nosplit.json > $R/gpurun_out/r7c_insitu_diff_nosplit.txt 2>&1
cat $R/gpurun_out/r7c_insitu_diff_nosplit.txt | cut -c1-200
cd $R
ab() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $ARGS2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('several_clips_in_flight') or {}; print('$name', d['value'], d['ms_per_step'], 'several', s.get('value'))" >> gpurun_out/r7c_ab.txt
}
BASE="--config 3 --steps 274 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-full-session --no-sustained"
for i in 1 2; do
  ARGS2="$BASE --lanes 2"
  ab base_share_aware X=1
  ab shapeonly_full MIVOS_PP_SHARE_CAP=0
  ab shapeonly_min_nk64 MIVOS_PP_SHARE_CAP=0 MIVOS_PP_SPLIT_MIN_NK=64
  ab shapeonly_min_nk128 MIVOS_PP_SHARE_CAP=0 MIVOS_PP_SPLIT_MIN_NK=128
  ab shapeonly_min_nk256 MIVOS_PP_SHARE_CAP=0 MIVOS_PP_SPLIT_MIN_NK=256
  ab nosplit MIVOS_PP_SPLIT_THR=0
done
cat gpurun_out/r7c_ab.txt

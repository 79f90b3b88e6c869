#!/bin/bash
# round 6, call 5: folded split-K (bit-identical under chip_share) - tests, then same-box A/B against the round-5 tree with one / two / three clips in flight
# (configs 3 and 2), fusion beside memorize, and the driver's bench command
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -q -m gpu -k "chip_share or lanes or generator or lds_dma or fp16_range or end_to_end" > gpurun_out/r7e_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r7e_pytest.log | cut -c1-300
grep -E "^FAILED|^ERROR" gpurun_out/r7e_pytest.log | head -20
new() {  # name, config args..., then env after --
  name=$1; shift; args=""; while [ "$1" != "--" ]; do args="$args $1"; shift; done; shift
  env "$@" timeout 300 python bench.py $args --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-full-session --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name: one clip', d['value'], 'several', (d.get('several_clips_in_flight') or {}).get('value'))" >> gpurun_out/r7e_ab.txt
}
old() {  # name, args
  name=$1; shift
  (cd build/r5tree && timeout 300 python bench.py "$@" --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-full-session 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name: one clip', (d.get('one_clip_in_flight') or {}).get('value'), 'several', d['value'])") >> gpurun_out/r7e_ab.txt
}
C3="--config 3 --steps 274 --warmup 137 --lanes 2"
C2="--config 2 --steps 276 --warmup 69 --lanes 3"
for i in 1 2; do
  old "r5tree config3 lanes2" $C3
  new "this   config3 lanes2 (fold)" $C3 -- X=1
  new "this   config3 lanes2 split (MIVOS_PP_FOLD=0)" $C3 -- MIVOS_PP_FOLD=0
  new "this   config3 lanes2 fuse beside memorize" $C3 -- MIVOS_FUSE_BESIDE_MEMORIZE=1
  old "r5tree config2 lanes3" $C2
  new "this   config2 lanes3 (fold)" $C2 -- X=1
  new "this   config2 lanes3 split (MIVOS_PP_FOLD=0)" $C2 -- MIVOS_PP_FOLD=0
done
cat gpurun_out/r7e_ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-other-configs > gpurun_out/r7e_bench_driverflags.json 2> gpurun_out/r7e_bench_driverflags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7e_bench_driverflags.json').read().strip().splitlines()[-1])
print('driver flags:', d['value'], d['ms_per_step'], 'several', (d.get('several_clips_in_flight') or {}).get('value'), 'full', d['full_session']['value'], (d['full_session'].get('several_clips_in_flight') or {}).get('value'),
      'sustained', d['sustained']['value'], d['sustained']['several_clips_in_flight']['value'], 'hbm', d['hbm_peak_allocated_gb'], 'roof', d['roofline']['frac'], d['roofline']['timed_region']['frac'], 'parity', d['parity']['min_iou_engine_vs_ref_fp32'], d['parity']['fp64']['gate_passed'])
PY

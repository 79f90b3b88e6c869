#!/bin/bash
# round 5, call 4: new GPU tests (GUI replay under autocast, lanes / passes determinism, loaders' resize mode, training hooks), the UNCHANGED reference entry script on
# the engine (reference tree shipped through the untracked gpurun_in/ scratch), the long session on the re-conditioned fixture (f16x3 and exact fp32), small A/Bs,
# one full default bench line
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
timeout 900 python -m pytest -q "tests/test_gpu_engine.py::test_gui_call_pattern_under_autocast_golden" "tests/test_gpu_engine.py::test_concurrent_passes_and_suite_lanes_are_bit_identical" tests/test_dataset.py tests/test_gpu_train.py -m gpu > gpurun_out/r6d_pytest.log 2>&1
echo "pytest rc $?"; grep -E "^event|passed|failed" gpurun_out/r6d_pytest.log | tail -20 | cut -c1-300
el tests
if [ -f gpurun_in/reference/eval_interactive_davis.py ]; then
  rm -f gpurun_out/entry_script_parity.jsonl
  MIVOS_REFERENCE_ROOT=$PWD/gpurun_in/reference timeout 900 python -m pytest -q -rs tests/test_entry_script.py -m gpu > gpurun_out/r6d_pytest_entry_script.log 2>&1
  echo "entry script pytest rc $?"; tail -5 gpurun_out/r6d_pytest_entry_script.log | cut -c1-300
  cp gpurun_out/entry_script_parity.jsonl gpurun_out/r6d_entry_script_parity.jsonl 2>/dev/null
fi
el entry
if [ -f gpurun_in/L32/done ]; then
  for prec in f16x3 f32; do
    timeout 400 python scripts/long_session_parity.py engine --precision $prec --mask-gain 0.3 --logit-gain 0.6 --ref32 gpurun_in/L32 $( [ -f gpurun_in/L64/done ] && echo "--ref64 gpurun_in/L64" ) --wait 5 \
      --json gpurun_out/r6d_long_session_parity_$prec.json 2>&1 | grep '^{' | python -c "
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
  echo "$name | 2 sessions: $b | t=$(( $(date +%s) - t0 ))" | tee -a gpurun_out/r6d_ab.txt
}
rm -f gpurun_out/r6d_ab.txt
ab default X=1
ab no_fuse_side_stream MIVOS_FUSE_SIDE_STREAM=0
ab qbatch20 MIVOS_QUERY_BATCH=20
ab qbatch5 MIVOS_QUERY_BATCH=5
ab default_b X=1
el ab
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6d_bench_driverflags.json 2> gpurun_out/r6d_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r6d_bench_driverflags.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], 'one lane', d.get('one_clip_in_flight'), 'full', {k: d['full_session'].get(k) for k in ('value', 'sessions_in_flight')}, (d['full_session'].get('one_clip_in_flight') or {}).get('value'))
print('roof', d['roofline'].get('frac'), d['roofline'].get('timed_region'), 'aff', d['roofline'].get('affinity', {}).get('frac'))
print('parity', {k: v for k, v in d.get('parity', {}).items() if k != 'fp64'}, {k: v for k, v in d.get('parity', {}).get('fp64', {}).items() if not k.startswith('per_frame')})
print('cpu', d.get('cpu_baseline'))"
tail -3 gpurun_out/r6d_bench.err
el bench
echo "total $(( $(date +%s) - t0 )) s"

#!/bin/bash
# round 6, call 4: whole GPU suite on the shape-only split-K rule (+ the first long-horizon fixture); same-box A/B of the round-5 tree (build/r5tree,
# git archive of 35462c6) against this tree with one and two clips in flight; the driver's bench command
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
rm -f gpurun_out/long_horizon_parity.jsonl gpurun_out/parity_ratios.jsonl
timeout 1800 python -m pytest tests -q -m gpu --deselect tests/test_gpu_long_horizon.py::test_long_horizon_fixtures_are_committed > gpurun_out/r7d_pytest.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/r7d_pytest.log | cut -c1-300
grep -E "^FAILED|^ERROR" gpurun_out/r7d_pytest.log | head -20
cat gpurun_out/long_horizon_parity.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    for it in d['interactions']:
        print(d['fixture'], 'admitted', d['admitted'], 'interact', it['interact'], 'min iou', round(it['min_iou'], 6), 'ref self', round(it['reference_self_min_iou'], 6), 'below', it['frames_below_bar'], 'e/r med', round(it['median_e_over_r'], 3), 'worst', round(it['worst_e_over_r'], 3), 'max e', it['max_e'], 'max r', it['max_r'], 'gate fail', it['gate_failures'])
"
A="--config 3 --steps 274 --warmup 137 --lanes 2 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-full-session"
for i in 1 2; do
  (cd build/r5tree && timeout 300 python bench.py $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r5tree  two lanes', d['value'], 'one lane', (d.get('one_clip_in_flight') or {}).get('value'))") >> gpurun_out/r7d_tree_ab.txt
  timeout 300 python bench.py $A --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('this tree two lanes', (d.get('several_clips_in_flight') or {}).get('value'), 'one lane', d['value'])" >> gpurun_out/r7d_tree_ab.txt
done
cat gpurun_out/r7d_tree_ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r7d_bench_driverflags.json 2> gpurun_out/r7d_bench_driverflags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7d_bench_driverflags.json').read().strip().splitlines()[-1])
print('driver flags:', d['value'], d['ms_per_step'], 'several', (d.get('several_clips_in_flight') or {}).get('value'), 'full', d['full_session']['value'], (d['full_session'].get('several_clips_in_flight') or {}).get('value'),
      'sustained', d['sustained']['value'], d['sustained']['several_clips_in_flight']['value'], d['sustained'].get('roofline_timed_region'), 'hbm', d['hbm_peak_allocated_gb'], 'roof', d['roofline']['frac'], d['roofline']['timed_region']['frac'])
print('other', json.dumps(d.get('other_configs'))[:1500])
PY

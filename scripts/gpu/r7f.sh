#!/bin/bash
# round 6, call 6: fold-mode test + long-horizon replays (two fixtures); A/B of the independent-branch side stream (MIVOS_BRANCH_STREAM) with one / two clips in flight;
# config 4 (48 clips) against the round-5 tree on the same box
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
rm -f gpurun_out/long_horizon_parity.jsonl
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_long_horizon.py -q -m gpu -k "chip_share or lanes or long or full_session" --deselect tests/test_gpu_long_horizon.py::test_long_horizon_fixtures_are_committed > gpurun_out/r7f_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r7f_pytest.log | cut -c1-300
grep -E "^FAILED|^ERROR" gpurun_out/r7f_pytest.log | head -20
python - <<'PY'
import json
for l in open('gpurun_out/long_horizon_parity.jsonl'):
    d = json.loads(l)
    for it in d['interactions']:
        print(d['fixture'], 'admitted', d['admitted'], 'interact', it['interact'], 'min iou', round(it['min_iou'], 6), 'ref self', round(it['reference_self_min_iou'], 6), 'below', it['frames_below_bar'], 'e/r med', round(it['median_e_over_r'], 3), 'worst', round(it['worst_e_over_r'], 3), 'gate fail', it['gate_failures'])
PY
new() {  # name, config args..., then env after --
  name=$1; shift; args=""; while [ "$1" != "--" ]; do args="$args $1"; shift; done; shift
  env "$@" timeout 300 python bench.py $args --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-full-session --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name: one clip', d['value'], 'several', (d.get('several_clips_in_flight') or {}).get('value'))" >> gpurun_out/r7f_ab.txt
}
C3="--config 3 --steps 274 --warmup 137 --lanes 2"
for i in 1 2 3; do
  new "base" $C3 -- X=1
  new "branch stream" $C3 -- MIVOS_BRANCH_STREAM=1
done
cat gpurun_out/r7f_ab.txt
for i in 1 2; do
  (cd build/r5tree && timeout 400 python bench.py --config 4 --clips 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r5tree config4 48 clips', d['value'], d['config']['suite_checksum'])") >> gpurun_out/r7f_config4_ab.txt
  timeout 400 python bench.py --config 4 --clips 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('this   config4 48 clips', d['value'], d['config']['suite_checksum'])" >> gpurun_out/r7f_config4_ab.txt
  MIVOS_BRANCH_STREAM=1 timeout 400 python bench.py --config 4 --clips 48 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('this   config4 48 clips branch stream', d['value'], d['config']['suite_checksum'])" >> gpurun_out/r7f_config4_ab.txt
done
cat gpurun_out/r7f_config4_ab.txt

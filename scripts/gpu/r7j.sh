#!/bin/bash
# round 6, call 11: scale / bias in the epilogues as one fma (current tree) or as multiply + add with contraction switched off (what the reference's batch_norm does) - which
# one tracks the reference more closely: AttentionReadNetwork golden, the three long-horizon replays, the headline closed-loop test, teacher-forced logits
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
for lib in fma nofuse; do
  if [ $lib = fma ]; then unset MIVOS_HIP_LIB; else export MIVOS_HIP_LIB=$R/build/$lib/mivos_amd/libmivos_hip.so; fi
  python scripts/studies/attn_golden_probe.py $R $R 2>&1 | grep max | head -1 | sed "s/^/[$lib] /"
  rm -f gpurun_out/long_horizon_parity.jsonl gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl
  timeout 900 python -m pytest tests/test_gpu_long_horizon.py tests/test_gpu_engine.py tests/test_gpu_teacher_forced.py -q -m gpu -k "full_session_replay or headline or attention_read or end_to_end or teacher_forced_session" > gpurun_out/r7j_pytest_$lib.log 2>&1
  echo "[$lib] pytest rc $?"; tail -2 gpurun_out/r7j_pytest_$lib.log | cut -c1-200
  cp gpurun_out/long_horizon_parity.jsonl gpurun_out/r7j_long_$lib.jsonl; cp gpurun_out/parity_ratios.jsonl gpurun_out/r7j_ratios_$lib.jsonl; cp gpurun_out/teacher_forced.jsonl gpurun_out/r7j_teacher_$lib.jsonl 2>/dev/null
  python - <<PY
import json
for l in open('gpurun_out/r7j_long_$lib.jsonl'):
    d = json.loads(l)
    for it in d['interactions']:
        print('[$lib]', d['fixture'], it['interact'], 'min iou', round(it['min_iou'], 6), 'mean', round(it['mean_iou'], 6), 'vs fp64', round(it['min_iou_vs_fp64'], 6), 'e/r med', round(it['median_e_over_r'], 3), 'worst', round(it['worst_e_over_r'], 3), 'max dprob', '%.2e' % it['max_dprob_vs_reference'])
for l in open('gpurun_out/r7j_ratios_$lib.jsonl'):
    d = json.loads(l)
    print('[$lib]', d.get('tag'), 'worst ratio', d.get('worst_ratio'), 'passed', d.get('passed'))
PY
done

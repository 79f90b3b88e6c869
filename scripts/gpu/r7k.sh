#!/bin/bash
# round 6, call 12: epilogue arithmetic pinned to round 5's (fused in the kernels, multiply + add in the reduce pass and in the folded kernel): the golden probe must read
# exactly what the round-5 tree reads; fold == split bitwise; then the whole GPU suite
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
python scripts/studies/attn_golden_probe.py $R $R 2>&1 | grep max | head -1
python scripts/studies/attn_golden_probe.py $R/build/r5tree $R 2>&1 | grep max | head -1
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl gpurun_out/long_horizon_parity.jsonl gpurun_out/entry_script_parity.jsonl
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r7k_pytest.log 2>&1
echo "full pytest rc $?"; tail -3 gpurun_out/r7k_pytest.log | cut -c1-250
grep -E "^FAILED|^ERROR" gpurun_out/r7k_pytest.log | head
python - <<'PY'
import json
for l in open('gpurun_out/long_horizon_parity.jsonl'):
    d = json.loads(l)
    for it in d['interactions']:
        print(d['fixture'], it['interact'], 'min iou', round(it['min_iou'], 6), 'mean', round(it['mean_iou'], 6), 'ref self', round(it['reference_self_min_iou'], 6), 'e/r med', round(it['median_e_over_r'], 3), 'worst', round(it['worst_e_over_r'], 3), 'max e', '%.2e' % it['max_e'], 'max r', '%.2e' % it['max_r'], 'mismatch px worst', it['mismatch_px_worst_frame'])
PY

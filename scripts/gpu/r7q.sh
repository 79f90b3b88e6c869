#!/bin/bash
# round 6: the three long-horizon sessions with the engine in its exact-fp32 mode (every convolution and the affinity on fp32 MFMA) - record only
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
rm -f gpurun_out/long_horizon_parity.jsonl
python - <<'PY'
import glob, json, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_long_horizon as T
for p in T.FIXTURES:
    rec = T.replay(p, precision="f32")
    for it in rec["interactions"]:
        print(rec["fixture"], "exact f32: interact", it["interact"], "min iou", round(it["min_iou"], 6), "mean", round(it["mean_iou"], 6), "e/r med", round(it["median_e_over_r"], 3), "worst", round(it["worst_e_over_r"], 3), "gate fail", it["gate_failures"])
PY
cp gpurun_out/long_horizon_parity.jsonl gpurun_out/r7q_long_horizon_exact_f32.jsonl

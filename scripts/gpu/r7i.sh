#!/bin/bash
# round 6, call 10: why test_attention_read_network_golden moved (explicit fma in the epilogues?) and why config 2 with three clips in flight reads low in other_configs
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
python scripts/studies/attn_golden_probe.py $R $R 2>&1 | grep max
MIVOS_HIP_LIB=$R/build/nofma/mivos_amd/libmivos_hip.so python scripts/studies/attn_golden_probe.py $R $R 2>&1 | grep max | sed 's/^/[nofma lib] /'
python scripts/studies/attn_golden_probe.py $R/build/r5tree $R 2>&1 | grep max
C2="--config 2 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-sustained --no-full-session --lanes 3"
for i in 1 2; do
  timeout 300 python bench.py $C2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 default steps: one clip', d['value'], 'several', (d.get('several_clips_in_flight') or {}).get('value'))"
  timeout 300 python bench.py $C2 --steps 276 --warmup 69 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 276 steps: one clip', d['value'], 'several', (d.get('several_clips_in_flight') or {}).get('value'))"
  MIVOS_BENCH_SAME_CLIP=1 timeout 300 python bench.py $C2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 default steps, same clip in every lane: one clip', d['value'], 'several', (d.get('several_clips_in_flight') or {}).get('value'))"
done

#!/bin/bash
# round 6, call 7: query-batch prefetch on a side stream (ordinary / lowest HIP priority): correctness under the env switch, then same-box A/B with one / two clips in flight
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
python - <<'PY'
import torch
from mivos_amd import ops
s = ops._low_priority_stream(torch.device("cuda:0"))
print("low-priority stream:", s, None if s is None else s.priority if hasattr(s, "priority") else "?")
PY
MIVOS_QUERY_PREFETCH=low timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "end_to_end or headline or lanes or interaction_order or 480p_propagation" > gpurun_out/r7g_pytest_prefetch.log 2>&1
echo "pytest (prefetch low) rc $?"; tail -3 gpurun_out/r7g_pytest_prefetch.log | cut -c1-300
new() {  # name, config args..., then env after --
  name=$1; shift; args=""; while [ "$1" != "--" ]; do args="$args $1"; shift; done; shift
  env "$@" timeout 300 python bench.py $args --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name: one clip', d['value'], 'several', (d.get('several_clips_in_flight') or {}).get('value'), 'full', d['full_session']['value'], (d['full_session'].get('several_clips_in_flight') or {}).get('value'))" >> gpurun_out/r7g_ab.txt
}
C3="--config 3 --steps 274 --warmup 137 --lanes 2"
for i in 1 2; do
  new "base" $C3 -- X=1
  new "prefetch side stream" $C3 -- MIVOS_QUERY_PREFETCH=1
  new "prefetch low priority" $C3 -- MIVOS_QUERY_PREFETCH=low
  new "prefetch low priority at 9" $C3 -- MIVOS_QUERY_PREFETCH=low MIVOS_QUERY_PREFETCH_AT=9
  new "prefetch low priority at 3" $C3 -- MIVOS_QUERY_PREFETCH=low MIVOS_QUERY_PREFETCH_AT=3
done
cat gpurun_out/r7g_ab.txt

#!/bin/bash
set +e
export TMPDIR=/tmp
for v in 7 21 0 7 21; do
timeout 100 python bench.py --steps 548 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 --profile-every $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline'] or {}; print('profile_every=$v', d['value'], d['ms_per_step'], r.get('launches_sampled'), (r.get('affinity') or {}).get('launches_sampled'))"
done

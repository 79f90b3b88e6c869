#!/bin/bash
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { name=$1; shift; timeout 300 python bench.py --config 2 --cpu-frames 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('several_clips_in_flight') or {}; print('$name: one clip', d['value'], 'several', s.get('value'), s.get('steps'), s.get('warmup'), 'full', (d.get('full_session') or {}).get('value'), ((d.get('full_session') or {}).get('several_clips_in_flight') or {}).get('value'), 'sustained', (d.get('sustained') or {}).get('value'), ((d.get('sustained') or {}).get('several_clips_in_flight') or {}).get('value'))"; }
run "default"
run "profile-every 0" --profile-every 0
run "no full/sustained" --no-full-session --no-sustained
run "profile-every 0, no full/sustained" --profile-every 0 --no-full-session --no-sustained
run "default again"

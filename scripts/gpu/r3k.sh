#!/bin/bash
set +e
export TMPDIR=/tmp
timeout 60 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('short', d['value'], r['launches_sampled'], r['frac'], r['affinity']['launches_sampled'])"
timeout 100 python bench.py --cpu-frames 0 --exact-f32-steps 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('default', d['value'], d['ms_per_step'], r['launches_sampled'], r['frac'], r['affinity']['launches_sampled'], r['affinity']['avg_launch_us'], r['traffic'] and r['traffic']['source'])"

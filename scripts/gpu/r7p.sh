#!/bin/bash
# round 6: rocprofv3 kernel stats with TWO sessions in flight (the folded split-K kernel and the fusion-beside-memorize schedule are what this mode runs)
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; rm -rf /tmp/ks2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks2 --output-format csv -- python $R/bench.py --config 3 --lanes 2 --steps 274 --warmup 137 --no-full-session --no-sustained --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > $R/gpurun_out/r7p_stats_bench.json 2> /tmp/ks2.err
cp "$(find /tmp/ks2 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r7p_config3_two_sessions_kernel_stats.csv
head -12 $R/gpurun_out/r7p_config3_two_sessions_kernel_stats.csv | cut -c1-140
python -c "
import json; d=json.loads(open('$R/gpurun_out/r7p_stats_bench.json').read().strip().splitlines()[-1]); print('under rocprof: one clip', d['value'], 'two clips', d['several_clips_in_flight']['value'])"

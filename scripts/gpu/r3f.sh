#!/bin/bash
# rocprofv3 kernel stats of bench config 5 (1080p x 1000 frames, 3 objects, bank to T=200)
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks5
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks5 --output-format csv -- python $R/bench.py --config 5 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > $R/gpurun_out/r02i_c5_stats_bench.json 2> /tmp/ks5.err
f=$(find /tmp/ks5 -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r02i_c5_kernel_stats.csv
head -12 $R/gpurun_out/r02i_c5_kernel_stats.csv | cut -c1-160
cut -c1-200 $R/gpurun_out/r02i_c5_stats_bench.json

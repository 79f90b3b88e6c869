#!/bin/bash
# rocprofv3 evidence for one bench command (run on the GPU box through gpurun):  scripts/profile_bench.sh <tag> [bench args...]
#   1. --kernel-trace --stats   -> gpurun_out/<tag>_kernel_stats.csv
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md: the two do not fit one pass; never combined
#      with trace domains) -> gpurun_out/<tag>_pmc_traffic.json via scripts/pmc_traffic.py (FETCH_SIZE x2, gfx950 correction)
set +e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
ARGS="$@ --cpu-frames 0 --exact-f32-steps 0 --profile-every 0"
rm -rf /tmp/ks /tmp/pm_FETCH_SIZE /tmp/pm_WRITE_SIZE
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python $R/bench.py $ARGS > $R/gpurun_out/${TAG}_stats_bench.json 2> /tmp/ks.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d /tmp/pm_$c --output-format csv -- python $R/bench.py $ARGS > /dev/null 2> /tmp/pm_$c.err
done
python $R/scripts/pmc_traffic.py $(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/${TAG}_pmc_traffic.json
head -14 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
cut -c1-200 $R/gpurun_out/${TAG}_stats_bench.json

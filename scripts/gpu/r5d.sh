#!/bin/bash
# round 4, call 4: long closed-loop session with fp64 arbitration (both CPU oracles computed beforehand, gpurun_in/), PMC passes of the
# config-3 bench (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes; MfmaUtil), config 2 line.
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
t0=$(date +%s)
timeout 600 python scripts/long_session_parity.py engine --ref32 gpurun_in/long32 --ref64 gpurun_in/long64 --wait 5 --json gpurun_out/r5d_long_session_parity.json > gpurun_out/r5d_long_engine.log 2>&1
tail -3 gpurun_out/r5d_long_engine.log | cut -c1-1800
echo "t=$(( $(date +%s) - t0 ))"
cd /tmp
ARGS="--cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 137 --warmup 8"
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rm -rf /tmp/pm_$c
  timeout 400 rocprofv3 --pmc $c -d /tmp/pm_$c --output-format csv -- python $R/bench.py $ARGS > /tmp/pm_$c.json 2> /tmp/pm_$c.err
  echo "$c rc $? t=$(( $(date +%s) - t0 ))"
done
python $R/scripts/pmc_traffic.py $(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/r5d_config3_pmc_traffic.json | head -12
python $R/scripts/pmc_mfma_util.py $(find /tmp/pm_MfmaUtil -name "*counter_collection.csv" | head -1) $R/gpurun_out/r5d_config3_mfma_util.json | head -14 | cut -c1-160
cd $R
timeout 200 python bench.py --config 2 --cpu-frames 0 2>/dev/null | tee gpurun_out/r5d_bench_config2.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2', d['value'], d['ms_per_step'], (d.get('full_session') or {}).get('value'))"
echo "total $(( $(date +%s) - t0 )) s"

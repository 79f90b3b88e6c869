#!/bin/bash
# round 4, call 6: memory-read select kernel with its chunks dealt XCD-major (MIVOS_SELECT_XCD=1): exact index tests, A/B, fabric traffic
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
t0=$(date +%s)
MIVOS_SELECT_XCD=1 timeout 500 python -m pytest -q -x tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -k "memory_read or split_keys or single_step or segment_with_query or end_to_end or fusion_net_forward or upsample" > gpurun_out/r5f_pytest.log 2>&1
echo "pytest rc $? after $(( $(date +%s) - t0 )) s"; tail -4 gpurun_out/r5f_pytest.log | cut -c1-300
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
run() {
  name=$1; shift
  a=$(env "$@" timeout 200 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  b=$(env "$@" timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$name | driver window: $a | 2 sessions: $b | t=$(( $(date +%s) - t0 ))" | tee -a gpurun_out/r5f_ab.txt
}
rm -f gpurun_out/r5f_ab.txt
run selxcd1 MIVOS_SELECT_XCD=1
run selxcd0 MIVOS_SELECT_XCD=0
run selxcd1 MIVOS_SELECT_XCD=1
run selxcd0 MIVOS_SELECT_XCD=0
run selxcd1 MIVOS_SELECT_XCD=1
run selxcd0 MIVOS_SELECT_XCD=0
cd /tmp
ARGS="--cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 137 --warmup 8"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  MIVOS_SELECT_XCD=1 timeout 300 rocprofv3 --pmc $c -d /tmp/pm_$c --output-format csv -- python $R/bench.py $ARGS > /tmp/pm_$c.json 2> /tmp/pm_$c.err
done
python $R/scripts/pmc_traffic.py $(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/r5f_config3_pmc_traffic_selxcd1.json | head -14
for v in 1 0; do
  rm -rf /tmp/ks$v
  MIVOS_SELECT_XCD=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks$v --output-format csv -- python $R/bench.py $ARGS > /dev/null 2> /tmp/ks$v.err
  f=$(find /tmp/ks$v -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r5f_config3_kernel_stats_selxcd$v.csv
done
echo "total $(( $(date +%s) - t0 )) s"

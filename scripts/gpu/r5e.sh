#!/bin/bash
# round 4, call 5: XCD-contiguous work assignment of the tile / element walking kernels (upsample, maxpool, stem, FusionNet conv1 / resblock /
# head, memread finalize): correctness on the kernels' own tests, A/B against the round-robin walk (MIVOS_XCD_CONTIG=0), fabric traffic, kernel times.
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
t0=$(date +%s)
timeout 600 python -m pytest -q -x tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -k "upsample or maxpool or fusion or stem or memory_read or golden or decoder or single_step or segment_with_query" > gpurun_out/r5e_pytest.log 2>&1
echo "pytest rc $? after $(( $(date +%s) - t0 )) s"; tail -4 gpurun_out/r5e_pytest.log | cut -c1-300
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
run() {
  name=$1; shift
  a=$(env "$@" timeout 200 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  b=$(env "$@" timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$name | driver window: $a | 2 sessions: $b | t=$(( $(date +%s) - t0 ))" | tee -a gpurun_out/r5e_ab.txt
}
rm -f gpurun_out/r5e_ab.txt
run contig1 MIVOS_XCD_CONTIG=1
run contig0 MIVOS_XCD_CONTIG=0
run contig1 MIVOS_XCD_CONTIG=1
run contig0 MIVOS_XCD_CONTIG=0
cd /tmp
ARGS="--cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 137 --warmup 8"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 300 rocprofv3 --pmc $c -d /tmp/pm_$c --output-format csv -- python $R/bench.py $ARGS > /tmp/pm_$c.json 2> /tmp/pm_$c.err
done
python $R/scripts/pmc_traffic.py $(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/r5e_config3_pmc_traffic.json | head -16
for v in 1 0; do
  rm -rf /tmp/ks$v
  MIVOS_XCD_CONTIG=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ks$v --output-format csv -- python $R/bench.py $ARGS > /dev/null 2> /tmp/ks$v.err
  f=$(find /tmp/ks$v -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r5e_config3_kernel_stats_contig$v.csv
  echo "== contig $v"; grep -E "upsample|maxpool|stem|fusion_|finalize" $R/gpurun_out/r5e_config3_kernel_stats_contig$v.csv | cut -d, -f1-4 | cut -c1-120
done
echo "total $(( $(date +%s) - t0 )) s"

#!/bin/bash
# round 6, call 1: Step A of the round-5 review - per-layer-shape IN-SITU kernel times (rocprofv3 kernel trace joined with the library's launch log) under the
# default tile rules and under the "small tiles, no split-K" rules that win 20-30 % per layer stand-alone; sustained clocks / power beside one and two clips in flight.
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
ARGS="--config 3 --lanes 1 --steps 274 --warmup 137 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 --no-full-session"
for tag in default small128; do
  rm -rf /tmp/kt_$tag /tmp/conv_$tag.log
  if [ $tag = small128 ]; then export MIVOS_PP_SMALL_WGS=128; else unset MIVOS_PP_SMALL_WGS; fi
  MIVOS_CONV_LOG=/tmp/conv_$tag.log timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_$tag --output-format csv -- python $R/bench.py $ARGS > $R/gpurun_out/r7a_trace_bench_$tag.json 2> /tmp/kt_$tag.err
  python $R/scripts/insitu_shape_table.py /tmp/kt_$tag /tmp/conv_$tag.log --label $tag --json $R/gpurun_out/r7a_insitu_$tag.json > $R/gpurun_out/r7a_insitu_$tag.txt 2>&1
  head -5 $R/gpurun_out/r7a_insitu_$tag.txt | cut -c1-250
done
unset MIVOS_PP_SMALL_WGS
python $R/scripts/insitu_shape_table.py --diff $R/gpurun_out/r7a_insitu_default.json $R/gpurun_out/r7a_insitu_small128.json > $R/gpurun_out/r7a_insitu_diff.txt 2>&1
cat $R/gpurun_out/r7a_insitu_diff.txt | cut -c1-200
f=$(find /tmp/kt_default -name "*kernel_trace.csv" | head -1); gzip -c "$f" > $R/gpurun_out/r7a_kernel_trace_default.csv.gz
cp /tmp/conv_default.log $R/gpurun_out/r7a_conv_default.log
# same box, no profiler: A/B/A/B of the two rule sets (one clip in flight, 3 sessions timed)
for i in 1 2; do for tag in default small128; do
  if [ $tag = small128 ]; then export MIVOS_PP_SMALL_WGS=128; else unset MIVOS_PP_SMALL_WGS; fi
  python $R/bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" >> $R/gpurun_out/r7a_rules_ab.txt
done; done
unset MIVOS_PP_SMALL_WGS
cat $R/gpurun_out/r7a_rules_ab.txt
# sustained clocks / power: rocm-smi sampled beside 8 sessions, one clip in flight and two
for lanes in 1 2; do
  ( while true; do echo "T $(date +%s.%N)"; rocm-smi --showclocks --showpower --json 2>/dev/null | head -c 2000; echo; sleep 0.25; done ) > $R/gpurun_out/r7a_smi_lanes$lanes.txt &
  SMI=$!
  python $R/bench.py --config 3 --lanes $lanes --cpu-frames 0 --exact-f32-steps 0 > $R/gpurun_out/r7a_bench_sustained_lanes$lanes.json 2>/dev/null
  kill $SMI
  cut -c1-300 $R/gpurun_out/r7a_bench_sustained_lanes$lanes.json
done
tail -c 1500 $R/gpurun_out/r7a_smi_lanes1.txt

#!/bin/bash
# round 6, final tree: rocprofv3 kernel stats + kernel trace (per-shape table, timeline) + PMC passes of the one-session run (the committed records carry this tree's csrc
# fingerprint), then the driver's bench command and a 2-rank gloo plumbing run of the new line on one GPU
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
ARGS="--config 3 --lanes 1 --steps 274 --warmup 137 --no-full-session --no-sustained"
bash scripts/profile_bench.sh r7y_config3 $ARGS > gpurun_out/r7y_profile.log 2>&1
tail -2 gpurun_out/r7y_profile.log | cut -c1-160
cd /tmp; rm -rf /tmp/mu /tmp/kt
timeout 600 rocprofv3 --pmc MfmaUtil -d /tmp/mu --output-format csv -- python $R/bench.py --config 3 --lanes 1 --steps 137 --warmup 137 --no-full-session --no-sustained --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > /dev/null 2> /tmp/mu.err
python $R/scripts/pmc_mfma_util.py "$(find /tmp/mu -name '*counter_collection.csv' | head -1)" $R/gpurun_out/r7y_config3_mfma_util.json | head -8 | cut -c1-170
MIVOS_CONV_LOG=/tmp/conv.log timeout 600 rocprofv3 --kernel-trace -d /tmp/kt --output-format csv -- python $R/bench.py $ARGS --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > /dev/null 2> /tmp/kt.err
python $R/scripts/insitu_shape_table.py /tmp/kt /tmp/conv.log --label final --json $R/gpurun_out/r7y_insitu.json > $R/gpurun_out/r7y_insitu.txt 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/scripts/trace_timeline.py "$f" > $R/gpurun_out/r7y_timeline.md 2>&1
cd $R
el profiles
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r7y_bench_config3_driverflags.json 2> gpurun_out/r7y_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7y_bench_config3_driverflags.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'several', (d.get('several_clips_in_flight') or {}).get('value'), 'full', d['full_session']['value'], (d['full_session'].get('several_clips_in_flight') or {}).get('value'),
      '\n sustained', d['sustained']['value'], d['sustained']['clocks'], d['sustained']['several_clips_in_flight']['value'], d['sustained'].get('roofline_timed_region'),
      '\n roof', r['kernel'], r['frac'], r.get('by_bounding_roofline'), 'timed', r['timed_region']['frac'], 'traffic', r.get('traffic'), 'mfma', r.get('mfma_util_pmc'), '\n aff', {k: r['affinity'].get(k) for k in ('frac', 'avg_launch_us', 'traffic')},
      '\n parity', d['parity']['min_iou_engine_vs_ref_fp32'], d['parity']['fp64']['gate_passed'], d['parity']['fp64']['worst_frame_ratio'], 'cpu', d['cpu_baseline']['value'], 'hbm', d['hbm_peak_allocated_gb'],
      '\n other', json.dumps(d.get('other_configs'))[:1500])
PY
el bench
MIVOS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 > gpurun_out/r7y_bench_2ranks_one_gpu_gloo_plumbing.json 2> gpurun_out/r7y_2ranks.err
python -c "
import json; d=json.loads(open('gpurun_out/r7y_bench_2ranks_one_gpu_gloo_plumbing.json').read().strip().splitlines()[-1])
print('2 ranks on one GPU (gloo plumbing):', d['value'], d['n_gpus'], d.get('dist_backend'), d.get('gloo_ranks'), d['per_rank'], 'sustained', d['sustained']['value'])" || tail -5 gpurun_out/r7y_2ranks.err
timeout 900 python bench.py --config 4 > gpurun_out/r7y_bench_config4_474clips.json 2> gpurun_out/r7y_config4.err
python -c "
import json; d=json.loads(open('gpurun_out/r7y_bench_config4_474clips.json').read().strip().splitlines()[-1])
print('config 4, 474 clips:', d['value'], d['steps'], d['wall_seconds'], d['config']['clips_in_flight_per_gpu'], d['config']['suite_checksum'])" || tail -3 gpurun_out/r7y_config4.err
timeout 900 python bench.py --no-other-configs > gpurun_out/r7y_bench_config3_default_flags.json 2> gpurun_out/r7y_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r7y_bench_config3_default_flags.json').read().strip().splitlines()[-1])
print('config 3, default flags (8 sessions):', d['value'], d['ms_per_step'], d['clocks'], 'several', d['several_clips_in_flight']['value'], 'roof', d['roofline']['frac'], d['roofline']['timed_region']['frac'], d['sustained'].get('roofline_timed_region'))" || tail -3 gpurun_out/r7y_default.err
echo "total $(( $(date +%s) - t0 )) s"

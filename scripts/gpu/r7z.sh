#!/bin/bash
# round 6, final tree: whole GPU suite (incl. the three long-horizon replays), rocprofv3 kernel stats + trace + PMC passes of the one-session run (the committed
# records must carry this tree's csrc fingerprint), the driver's bench command, the auxiliary bench modes
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl gpurun_out/long_horizon_parity.jsonl gpurun_out/entry_script_parity.jsonl
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/r7z_pytest.log 2>&1
echo "full pytest rc $?"; tail -3 gpurun_out/r7z_pytest.log | cut -c1-250
grep -E "^FAILED|^ERROR" gpurun_out/r7z_pytest.log | head
el tests
ARGS="--config 3 --lanes 1 --steps 274 --warmup 137 --no-full-session --no-sustained"
bash scripts/profile_bench.sh r7z_config3 $ARGS > gpurun_out/r7z_profile.log 2>&1
tail -3 gpurun_out/r7z_profile.log | cut -c1-160
cd /tmp; rm -rf /tmp/mu /tmp/kt
timeout 600 rocprofv3 --pmc MfmaUtil -d /tmp/mu --output-format csv -- python $R/bench.py --config 3 --lanes 1 --steps 137 --warmup 137 --no-full-session --no-sustained --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > /dev/null 2> /tmp/mu.err
python $R/scripts/pmc_mfma_util.py "$(find /tmp/mu -name '*counter_collection.csv' | head -1)" $R/gpurun_out/r7z_config3_mfma_util.json | head -8 | cut -c1-170
MIVOS_CONV_LOG=/tmp/conv.log timeout 600 rocprofv3 --kernel-trace -d /tmp/kt --output-format csv -- python $R/bench.py $ARGS --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > /dev/null 2> /tmp/kt.err
python $R/scripts/insitu_shape_table.py /tmp/kt /tmp/conv.log --label final --json $R/gpurun_out/r7z_insitu.json > $R/gpurun_out/r7z_insitu.txt 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python $R/scripts/trace_timeline.py "$f" > $R/gpurun_out/r7z_timeline.md 2>&1
head -16 $R/gpurun_out/r7z_timeline.md | cut -c1-160
cd $R
el profiles
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r7z_bench_config3_driverflags.json 2> gpurun_out/r7z_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7z_bench_config3_driverflags.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'several', (d.get('several_clips_in_flight') or {}).get('value'), 'full', d['full_session']['value'], (d['full_session'].get('several_clips_in_flight') or {}).get('value'),
      '\n sustained', {k: v for k, v in d['sustained'].items() if k not in ('note',)},
      '\n roof', r['kernel'], r['frac'], r.get('by_bounding_roofline'), 'timed', r['timed_region']['frac'], 'traffic', r.get('traffic'), 'mfma', r.get('mfma_util_pmc'), '\n aff', {k: r['affinity'].get(k) for k in ('frac', 'avg_launch_us', 'traffic')},
      '\n parity', d['parity']['min_iou_engine_vs_ref_fp32'], d['parity']['fp64']['gate_passed'], d['parity']['fp64']['worst_frame_ratio'], 'cpu', d['cpu_baseline']['value'], 'hbm', d['hbm_peak_allocated_gb'],
      '\n other', json.dumps(d.get('other_configs'))[:1200])
PY
el bench
for a in "--config train" "--config s2m" "--config 4 --generator --clips 8"; do
  timeout 400 python bench.py $a 2> gpurun_out/r7z_aux_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a ->', d['metric'][:60], d['value'], d['unit'], d.get('ms_per_step'))" || tail -5 gpurun_out/r7z_aux_err.txt
done
echo "total $(( $(date +%s) - t0 )) s"

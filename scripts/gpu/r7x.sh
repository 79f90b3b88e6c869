#!/bin/bash
# round 6, last call: what the driver runs at round end, in its order - smoke(), then the bench command (other configurations first, in their own processes)
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "[t=$(( $(date +%s) - t0 )) s] smoke"
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r7x_bench_config3_driverflags.json 2> gpurun_out/r7x_bench.err
echo "bench rc $? [t=$(( $(date +%s) - t0 )) s]"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r7x_bench_config3_driverflags.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'several', (d.get('several_clips_in_flight') or {}).get('value'), 'full', d['full_session']['value'], (d['full_session'].get('several_clips_in_flight') or {}).get('value'),
      '\n sustained', d['sustained']['value'], d['sustained']['clocks']['sclk_mhz_median'], d['sustained']['several_clips_in_flight']['value'], d['sustained'].get('roofline_timed_region', {}).get('frac_at_sampled_clock'),
      '\n roof', r['frac'], (r.get('by_bounding_roofline') or {}).get('roofline_time_over_measured_time'), 'traffic', r.get('traffic'), 'mfma', r.get('mfma_util_pmc'), '\n aff', {k: r['affinity'].get(k) for k in ('frac', 'avg_launch_us', 'traffic')},
      '\n parity', d['parity']['min_iou_engine_vs_ref_fp32'], d['parity']['fp64']['gate_passed'], 'cpu', d['cpu_baseline']['value'])
o=d.get('other_configs') or {}
for k,v in o.items(): print(' other', k, v.get('value'), (v.get('several_clips_in_flight') or {}).get('value'), (v.get('full_session') or {}).get('value'), v.get('error'))
PY

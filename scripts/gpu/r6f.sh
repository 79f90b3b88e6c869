#!/bin/bash
# round 5, call 6: the round's measurements - bench lines of every config, rocprofv3 kernel stats (two sessions in flight = the default command, and one), PMC traffic
# and MfmaUtil passes of config 3 stamped with this tree's csrc fingerprint
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
R=$PWD
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
for e in "X=1" "MIVOS_MEMREAD_WGS=256" "X=1"; do
  env $e timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'], (d.get('one_clip_in_flight') or {}).get('value'))" | tee -a gpurun_out/r6f_select_share_ab.txt
done
el ab
bash scripts/profile_bench.sh r6f_config3 --config 3 --steps 274 --warmup 137 --no-full-session > gpurun_out/r6f_profile.log 2>&1
tail -16 gpurun_out/r6f_profile.log | cut -c1-160
el profile_lanes2
cd /tmp
rm -rf /tmp/ks1 /tmp/mu
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks1 --output-format csv -- python $R/bench.py --config 3 --lanes 1 --steps 274 --warmup 137 --no-full-session --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > $R/gpurun_out/r6f_config3_lanes1_stats_bench.json 2> /tmp/ks1.err
cp "$(find /tmp/ks1 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r6f_config3_lanes1_kernel_stats.csv
timeout 600 rocprofv3 --pmc MfmaUtil -d /tmp/mu --output-format csv -- python $R/bench.py --config 3 --lanes 1 --steps 137 --warmup 137 --no-full-session --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > /dev/null 2> /tmp/mu.err
python $R/scripts/pmc_mfma_util.py "$(find /tmp/mu -name '*counter_collection.csv' | head -1)" $R/gpurun_out/r6f_config3_mfma_util.json | head -12 | cut -c1-170
cd $R
el profile_lanes1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6f_bench_config3_driverflags.json 2> gpurun_out/r6f_bench.err
timeout 900 python bench.py > gpurun_out/r6f_bench_config3.json 2>> gpurun_out/r6f_bench.err
el bench3
timeout 600 python bench.py --config 2 > gpurun_out/r6f_bench_config2.json 2>> gpurun_out/r6f_bench.err
el bench2
timeout 900 python bench.py --config 4 > gpurun_out/r6f_bench_config4_474clips.json 2>> gpurun_out/r6f_bench.err
el bench4
timeout 900 python bench.py --config 5 > gpurun_out/r6f_bench_config5.json 2>> gpurun_out/r6f_bench.err
el bench5
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6f_bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    r = d.get('roofline') or {}
    print(f.split('/')[-1], d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'lanes', d['config'].get('clips_in_flight_per_gpu'), 'one lane', (d.get('one_clip_in_flight') or {}).get('value'),
          'full', (d.get('full_session') or {}).get('value'), 'frac', r.get('frac'), (r.get('timed_region') or {}).get('frac'), 'aff', (r.get('affinity') or {}).get('frac'),
          'exact', (d.get('exact_f32') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', (d.get('parity') or {}).get('min_iou_engine_vs_ref_fp32'), ((d.get('parity') or {}).get('fp64') or {}).get('gate_passed'))
PY
echo "total $(( $(date +%s) - t0 )) s"

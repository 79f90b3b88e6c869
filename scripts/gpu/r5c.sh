#!/bin/bash
# round 4, final call: the driver-equivalent run - full GPU suite (records: parity_ratios / teacher_forced / entry_script), bench lines
# (driver flags, default), kernel stats under rocprofv3, config 5, config 4 (whole suite) if time is left.
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
t0=$(date +%s)
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl gpurun_out/entry_script_parity.jsonl
timeout 1300 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r5c_pytest.log 2>&1
echo "pytest rc $? after $(( $(date +%s) - t0 )) s"; tail -22 gpurun_out/r5c_pytest.log | cut -c1-300
python scripts/parity_clauses.py gpurun_out/parity_ratios.jsonl > gpurun_out/r5c_parity_clauses.txt 2>&1; tail -2 gpurun_out/r5c_parity_clauses.txt
R64=""; [ -f gpurun_in/long64/done ] && R64="--ref64 gpurun_in/long64"
timeout 600 python scripts/long_session_parity.py engine --ref32 gpurun_in/long32 $R64 --wait 5 --json gpurun_out/r5c_long_session_parity.json > gpurun_out/r5c_long_engine.log 2>&1
tail -3 gpurun_out/r5c_long_engine.log | cut -c1-1500
echo "t=$(( $(date +%s) - t0 ))"
sumline() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], 'full', (d.get('full_session') or {}).get('value'), r['kernel'], r['frac'], 'aff_us', r['affinity']['avg_launch_us'], r['affinity']['frac'], 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'parity', json.dumps(d.get('parity'))[:300])"; }
timeout 400 python bench.py --steps 20 --warmup 5 2> gpurun_out/r5c_bench_driver.err | tee gpurun_out/r5c_bench_config3_driverflags.json | sumline driverflags
timeout 400 python bench.py --cpu-frames 0 2> gpurun_out/r5c_bench.err | tee gpurun_out/r5c_bench_config3.json | sumline default8
echo "t=$(( $(date +%s) - t0 ))"
cd /tmp; rm -rf /tmp/ks
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python $R/bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 274 > $R/gpurun_out/r5c_stats_bench.json 2> /tmp/ks.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r5c_config3_kernel_stats.csv; head -12 $R/gpurun_out/r5c_config3_kernel_stats.csv | cut -c1-150
cd $R
echo "t=$(( $(date +%s) - t0 ))"
timeout 300 python bench.py --config 5 --cpu-frames 0 2> gpurun_out/r5c_bench5.err | tee gpurun_out/r5c_bench_config5.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['roofline']['affinity']; print('config5', d['value'], d['ms_per_step'], a['avg_launch_us'], a['frac'])"
echo "t=$(( $(date +%s) - t0 ))"
if [ $(( $(date +%s) - t0 )) -lt 1350 ]; then
  timeout 420 python bench.py --config 4 2> gpurun_out/r5c_bench4.err | tee gpurun_out/r5c_bench_config4.json | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['value'], d['ms_per_step'], d['steps'], d['config']['clips'])"
fi
echo "total $(( $(date +%s) - t0 )) s"

#!/bin/bash
# round 4, call 1: full GPU suite with the new parity tests (records: parity_ratios / teacher_forced / entry_script), the long closed-loop
# session (CPU oracles fp32 + fp64 in the background from the start), bench lines, Winograd proxy, kernel stats.
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl gpurun_out/entry_script_parity.jsonl
nohup python scripts/long_session_parity.py oracle --dtype fp32 --out /tmp/long32 --threads 48 > gpurun_out/r5a_oracle32.log 2>&1 &
nohup python scripts/long_session_parity.py oracle --dtype fp64 --out /tmp/long64 --threads 96 > gpurun_out/r5a_oracle64.log 2>&1 &
t0=$(date +%s)
timeout 1700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r5a_pytest.log 2>&1
echo "pytest rc $? after $(( $(date +%s) - t0 )) s"; tail -25 gpurun_out/r5a_pytest.log | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/r5a_pytest.log | cut -c1-300 | head -20
python scripts/parity_clauses.py gpurun_out/parity_ratios.jsonl > gpurun_out/r5a_parity_clauses.txt 2>&1; tail -4 gpurun_out/r5a_parity_clauses.txt
timeout 120 python scripts/studies/winograd_proxy.py > gpurun_out/r5a_winograd_proxy.json 2> gpurun_out/r5a_winograd_proxy.err; cat gpurun_out/r5a_winograd_proxy.json; tail -2 gpurun_out/r5a_winograd_proxy.err
sumline() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], 'full', (d.get('full_session') or {}).get('value'), r['kernel'], r['frac'], 'aff_us', r['affinity']['avg_launch_us'], 'parity', json.dumps(d.get('parity'))[:400])"; }
timeout 400 python bench.py --steps 20 --warmup 5 2> gpurun_out/r5a_bench_driver.err | tee gpurun_out/r5a_bench_config3_driverflags.json | sumline driverflags
timeout 400 python bench.py --cpu-frames 0 --exact-f32-steps 0 2> gpurun_out/r5a_bench.err | tee gpurun_out/r5a_bench_config3.json | sumline default8
echo "oracle logs:"; tail -2 gpurun_out/r5a_oracle32.log gpurun_out/r5a_oracle64.log
timeout 1500 python scripts/long_session_parity.py engine --ref32 /tmp/long32 --ref64 /tmp/long64 --wait 900 --json gpurun_out/r5a_long_session_parity.json > gpurun_out/r5a_long_engine.log 2>&1
tail -4 gpurun_out/r5a_long_engine.log | cut -c1-1200
cd /tmp; rm -rf /tmp/ks
timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python $R/bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 274 > $R/gpurun_out/r5a_stats_bench.json 2> /tmp/ks.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r5a_config3_kernel_stats.csv; head -14 $R/gpurun_out/r5a_config3_kernel_stats.csv | cut -c1-160
echo "total $(( $(date +%s) - t0 )) s"

#!/bin/bash
# round 3, final call: whole GPU test-suite, every bench line, rocprofv3 kernel stats + PMC traffic for configs 3 and 5
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
rm -f gpurun_out/parity_ratios.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=12 > gpurun_out/r4i_pytest.log 2>&1
echo "pytest rc $?"; tail -16 gpurun_out/r4i_pytest.log | cut -c1-200
timeout 500 python bench.py > gpurun_out/r4i_bench_config3.json 2> gpurun_out/r4i_bench_config3.err; echo "bench3 rc $?"
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r4i_bench_config3_driverflags.json 2>/dev/null; echo "bench3 driver rc $?"
timeout 300 python bench.py --config 2 > gpurun_out/r4i_bench_config2.json 2>/dev/null; echo "bench2 rc $?"
timeout 600 python bench.py --config 5 > gpurun_out/r4i_bench_config5.json 2> gpurun_out/r4i_bench_config5.err; echo "bench5 rc $?"
timeout 300 python bench.py --config 4 --clips 48 > gpurun_out/r4i_bench_config4_48clips.json 2>/dev/null; echo "bench4 rc $?"
timeout 300 python bench.py --config 4 --clips 8 --generator > gpurun_out/r4i_bench_generator_8clips.json 2>/dev/null; echo "gen rc $?"
timeout 200 python bench.py --config s2m > gpurun_out/r4i_bench_s2m.json 2>/dev/null; echo "s2m rc $?"
timeout 200 python bench.py --config train > gpurun_out/r4i_bench_train.json 2>/dev/null; echo "train rc $?"
for f in config3 config3_driverflags config2 config5 config4_48clips; do python - <<PY
import json
d=json.loads(open('gpurun_out/r4i_bench_$f.json').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print('$f', d['value'], d['ms_per_step'], d.get('full_session') and d['full_session']['value'], r.get('kernel'), r.get('frac'), (r.get('affinity') or {}).get('avg_launch_us'), (r.get('affinity') or {}).get('frac_of_f32_mfma_peak'), (d.get('parity') or {}).get('fp64', {}).get('gate_passed'))
PY
done
bash scripts/profile_bench.sh r4i_config3 --config 3 --no-full-session > gpurun_out/r4i_profile_config3.log 2>&1; tail -16 gpurun_out/r4i_profile_config3.log | cut -c1-160
bash scripts/profile_bench.sh r4i_config5 --config 5 --frames 400 > gpurun_out/r4i_profile_config5.log 2>&1; tail -12 gpurun_out/r4i_profile_config5.log | cut -c1-160

#!/bin/bash
# round 4, last call: the final defaults (QUERY_BATCH 10, XCD-contiguous pointwise walks + head, select chunks XCD-major): bench lines for the
# record first, then the GPU suite without its two slowest fp64 sessions (both green on the same kernels in call 3)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
t0=$(date +%s)
sumline() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], 'full', (d.get('full_session') or {}).get('value'), r['kernel'], r['frac'], 'aff_us', r['affinity']['avg_launch_us'], r['affinity']['frac'], 'traffic', (r['affinity'].get('traffic') or {}).get('read'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; }
timeout 300 python bench.py --steps 20 --warmup 5 2> gpurun_out/r5g_bench_driver.err | tee gpurun_out/r5g_bench_config3_driverflags.json | sumline driverflags
timeout 300 python bench.py --cpu-frames 0 2> gpurun_out/r5g_bench.err | tee gpurun_out/r5g_bench_config3.json | sumline default8
echo "t=$(( $(date +%s) - t0 ))"
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl gpurun_out/entry_script_parity.jsonl
timeout 420 python -m pytest tests -m gpu -q --deselect "tests/test_gpu_engine.py::test_headline_config_parity_with_fp64_arbitration[5-50-8]" --deselect tests/test_gpu_engine.py::test_1080p_three_objects_with_fusion_vs_oracle > gpurun_out/r5g_pytest.log 2>&1
echo "pytest rc $? t=$(( $(date +%s) - t0 ))"; tail -5 gpurun_out/r5g_pytest.log | cut -c1-300
python scripts/parity_clauses.py gpurun_out/parity_ratios.jsonl > gpurun_out/r5g_parity_clauses.txt 2>&1; tail -2 gpurun_out/r5g_parity_clauses.txt
echo "total $(( $(date +%s) - t0 )) s"

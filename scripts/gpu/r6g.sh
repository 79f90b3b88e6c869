#!/bin/bash
# round 5, call 7: the whole GPU suite on the final tree (gate factor 1.5), smoke(), and the multi-rank plumbing of bench.py with sessions in flight (2 ranks on one GPU over gloo)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl
timeout 1500 python -m pytest tests -q -m gpu --durations=6 > gpurun_out/r6g_pytest.log 2>&1
echo "pytest rc $? after $(( $(date +%s) - t0 )) s"; tail -14 gpurun_out/r6g_pytest.log | cut -c1-250
cp gpurun_out/parity_ratios.jsonl gpurun_out/r6g_parity_ratios.jsonl 2>/dev/null
cp gpurun_out/teacher_forced.jsonl gpurun_out/r6g_teacher_forced.jsonl 2>/dev/null
python scripts/parity_clauses.py gpurun_out/r6g_parity_ratios.jsonl > gpurun_out/r6g_parity_clauses.txt 2>&1; tail -4 gpurun_out/r6g_parity_clauses.txt | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2
MIVOS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 > gpurun_out/r6g_bench_config3_2ranks_one_gpu_gloo_plumbing.json 2> gpurun_out/r6g_2ranks.err
python -c "
import json; d=json.loads(open('gpurun_out/r6g_bench_config3_2ranks_one_gpu_gloo_plumbing.json').read().strip().splitlines()[-1])
print('2 ranks on one GPU (gloo, plumbing):', d['value'], d['n_gpus'], d['config']['clips_in_flight_per_gpu'], d.get('dist_backend'), d.get('rccl_ranks'), d['per_rank'])" ; tail -2 gpurun_out/r6g_2ranks.err
MIVOS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --config 4 --clips 24 > gpurun_out/r6g_bench_config4_2ranks_one_gpu_gloo_plumbing.json 2>> gpurun_out/r6g_2ranks.err
python -c "
import json; d=json.loads(open('gpurun_out/r6g_bench_config4_2ranks_one_gpu_gloo_plumbing.json').read().strip().splitlines()[-1])
print('config 4, 2 ranks on one GPU (gloo, plumbing):', d['value'], d['n_gpus'], d['config']['clips_in_flight_per_gpu'], d['per_rank'])"
echo "total $(( $(date +%s) - t0 )) s"

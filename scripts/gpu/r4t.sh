#!/bin/bash
# round 3 final: bench lines of configs 3 and 5 with the CPU baseline + fp64 parity, PMC traffic of the 256-query select kernel, small engine check
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 170 python bench.py 2>/dev/null | tail -1 > gpurun_out/r4t_bench_config3.json
python -c "
import json
d=json.loads(open('gpurun_out/r4t_bench_config3.json').read()); r=d['roofline']; print('config3', d['value'], d['ms_per_step'], d.get('full_session',{}).get('value'), r['affinity']['avg_launch_us'], r['affinity']['frac_of_f32_mfma_peak'], r['frac'], d['parity'].get('fp64',{}).get('gate_passed'), d['parity'].get('mean_iou_engine_vs_ref_fp32'))" | cut -c1-400
timeout 150 python bench.py --config 5 2>/dev/null | tail -1 > gpurun_out/r4t_bench_config5.json
python -c "
import json
d=json.loads(open('gpurun_out/r4t_bench_config5.json').read()); r=d['roofline']; print('config5', d['value'], d['ms_per_step'], r['affinity']['avg_launch_us'], r['affinity']['frac_of_f32_mfma_peak'], d['parity'].get('fp64',{}).get('gate_passed'))" | cut -c1-400
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc_$c
  timeout 60 rocprofv3 --pmc $c -d /tmp/pc_$c --output-format csv -- python $R/scripts/memread_case.py 3 100 8160 50 q256 > /dev/null 2> /tmp/pc_$c.err
  echo "case pmc $c rc $?"
done
python $R/scripts/pmc_traffic.py $(find /tmp/pc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/r4t_config5_memread256_T100_pmc_traffic.json | head -4
cd $R
timeout 100 python -m pytest tests/test_gpu_engine.py::test_end_to_end_golden tests/test_gpu_ops.py::test_memory_read_golden -m gpu -q 2>&1 | tail -2 | cut -c1-200

#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 600 python -m pytest "tests/test_gpu_engine.py::test_480p_propagation_vs_oracle[1]" tests/test_gpu_engine.py::test_fusion_generator_golden tests/test_gpu_engine.py::test_end_to_end_golden tests/test_gpu_engine.py::test_interaction_order_and_reinteraction_vs_oracle -m gpu -q -rP 2>&1 | grep -E "clauses|passed|failed" | cut -c1-330
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rP -k "stem" 2>&1 | grep -E "^stem|passed|failed" | cut -c1-200
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 600 rocprofv3 --pmc $c -d /tmp/pm_$c --output-format csv -- python $R/bench.py --config 5 --frames 262 --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > /dev/null 2> /tmp/pm_$c.err
  echo "pmc $c rc $?"; tail -2 /tmp/pm_$c.err | cut -c1-200
done
python $R/scripts/pmc_traffic.py $(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/r4k_config5_pmc_traffic.json | head -8
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc_$c
  timeout 300 rocprofv3 --pmc $c -d /tmp/pc_$c --output-format csv -- python $R/scripts/memread_case.py 3 100 8160 50 q128 > /dev/null 2> /tmp/pc_$c.err
  echo "case pmc $c rc $?"
done
python $R/scripts/pmc_traffic.py $(find /tmp/pc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/r4k_memread_case_T100_pmc_traffic.json | head -6

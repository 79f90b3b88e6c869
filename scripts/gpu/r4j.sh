#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
T='tests/test_gpu_engine.py::test_480p_propagation_vs_oracle[1]'
G='tests/test_gpu_engine.py::test_fusion_generator_golden'
for env in "MIVOS_STEM_PLANES=1" "MIVOS_STEM_PLANES=0" "MIVOS_FUSE_SIDE_STREAM=0"; do
  echo "== $env alone"
  env $env timeout 300 python -m pytest "$T" -m gpu -q -rP 2>&1 | grep -E "480p_closed_loop|passed|failed" | cut -c1-260
done
echo "== after the generator test (default env)"
timeout 300 python -m pytest "$G" "$T" -m gpu -q -rP 2>&1 | grep -E "480p_closed_loop|passed|failed" | cut -c1-260
echo "== whole engine file up to it"
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -rP -k "golden or get_W or segment_with_query or fusion_net_forward or fusion_generator or single_step or (480p_propagation_vs_oracle and not 3)" 2>&1 | grep -E "480p_closed_loop|passed|failed" | cut -c1-260

#!/bin/bash
# round 5, call 1: (1) how the fp16 matrix pipe rounds (ubench), (2) the two-stream passes / two-lane suite are bit-identical to the sequential order,
# the merged-half-step and 64x128 instantiations of the LDS-DMA convolution pass the bit-exact convolution tests, (3) per-shape tile / split-K sweep of
# the <128,128> population, (4) A/B: suite lanes 1/2/3 (config 4, 48 clips), two-stream passes on a mid-clip interaction, (5) the driver's bench line,
# (6) where the engine-vs-fp64 distance of the default precision comes from (affinity / SH32 path / stem kernel / Cout=1 projection switched one by one)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_f16_rounding.hip -o /tmp/mfr 2>/dev/null && /tmp/mfr > gpurun_out/r6a_mfma_rounding.txt 2>&1
cat gpurun_out/r6a_mfma_rounding.txt | cut -c1-250
el ubench
timeout 500 python -m pytest -q -x "tests/test_gpu_engine.py::test_concurrent_passes_and_suite_lanes_are_bit_identical" "tests/test_gpu_engine.py::test_end_to_end_golden" "tests/test_gpu_engine.py::test_interaction_order_and_reinteraction_vs_oracle" -m gpu > gpurun_out/r6a_pytest_conc.log 2>&1
echo "pytest conc rc $?"; tail -5 gpurun_out/r6a_pytest_conc.log | cut -c1-400
el conc
MIVOS_PP_MERGE=1 timeout 400 python -m pytest -q tests/test_gpu_ops.py -m gpu -k "conv2d or sh32 or dma" > gpurun_out/r6a_pytest_merge.log 2>&1
echo "pytest merge rc $?"; tail -3 gpurun_out/r6a_pytest_merge.log | cut -c1-400
MIVOS_PP_TILE=24 timeout 400 python -m pytest -q tests/test_gpu_ops.py -m gpu -k "conv2d or sh32 or dma" > gpurun_out/r6a_pytest_tile24.log 2>&1
echo "pytest tile24 rc $?"; tail -3 gpurun_out/r6a_pytest_tile24.log | cut -c1-400
el convtests
timeout 700 python scripts/conv_shape_sweep.py --json gpurun_out/r6a_conv_sweep.json > gpurun_out/r6a_conv_sweep.txt 2>&1
cat gpurun_out/r6a_conv_sweep.txt | cut -c1-330
el sweep
for L in 1 2 3 1 2; do
  timeout 300 python bench.py --config 4 --clips 48 --lanes $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config4 48 clips lanes', d['config'].get('clips_in_flight_per_gpu'), d['value'], 'frames/s checksum', d['config']['suite_checksum'])" | tee -a gpurun_out/r6a_lanes_ab.txt
done
el lanes
timeout 300 python scripts/midclip_bench.py --reps 3 2>&1 | tail -1 | tee gpurun_out/r6a_midclip_ab.json | cut -c1-600
el midclip
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r6a_bench_driverflags.json 2> gpurun_out/r6a_bench.err
python -c "import json; d=json.loads(open('gpurun_out/r6a_bench_driverflags.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], 'full', d.get('full_session'), 'roof', d['roofline'].get('frac'), d['roofline'].get('affinity', {}).get('frac'), 'parity', d.get('parity', {}).get('iou'))"
el bench
if [ -f gpurun_in/d32/done ]; then
  D="--frames 24 --clip-frames 70 --ref32 gpurun_in/d32 --ref64 gpurun_in/d64 --wait 5"
  run() { name=$1; shift; timeout 200 python scripts/long_session_parity.py engine $D "$@" --json gpurun_out/r6a_diag_$name.json 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$name', 'interact', d['interact'], 'e/r max', d.get('median_ratio_of_maxima'), 'q999', d.get('median_ratio_of_q999'), 'min IoU', d['min_iou'], 'ref32-vs-64 min IoU', d.get('min_iou_ref32_vs_fp64'))" | tee -a gpurun_out/r6a_diag.txt; }
  run default
  run affinity_f32 --affinity f32
  run no_act_path --no-act-path
  run no_stem --no-stem-kernel
  run no_proj --no-cout1-projection
  run exact_f32 --precision f32
fi
el diag
echo "total $(( $(date +%s) - t0 )) s"

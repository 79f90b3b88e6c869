#!/bin/bash
# round 5, call 8: the fp16-range guard of the convolution epilogues (new csrc): its test + the convolution / engine tests, same-box A/B against the previous
# library for speed, then the PMC passes and the kernel stats again (the committed records must carry this tree's csrc fingerprint), then the whole GPU suite
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
R=$PWD
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
timeout 600 python -m pytest -q -x tests/test_gpu_engine.py -m gpu -k "overflow or end_to_end_golden or concurrent or gui" > gpurun_out/r6h_pytest_guard.log 2>&1
echo "guard pytest rc $?"; tail -4 gpurun_out/r6h_pytest_guard.log | cut -c1-300
timeout 600 python -m pytest -q tests/test_gpu_ops.py -m gpu -k "conv2d or sh32 or dma" > gpurun_out/r6h_pytest_conv.log 2>&1
echo "conv pytest rc $?"; tail -2 gpurun_out/r6h_pytest_conv.log | cut -c1-300
el tests
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
for i in 1 2; do
  timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with guard', d['value'], (d.get('one_clip_in_flight') or {}).get('value'))" | tee -a gpurun_out/r6h_guard_ab.txt
  if [ -f gpurun_in/libmivos_hip_before_guard.so ]; then
    MIVOS_HIP_LIB=$PWD/gpurun_in/libmivos_hip_before_guard.so timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('before guard', d['value'], (d.get('one_clip_in_flight') or {}).get('value'))" | tee -a gpurun_out/r6h_guard_ab.txt
  fi
done
el ab
bash scripts/profile_bench.sh r6h_config3 --config 3 --steps 274 --warmup 137 --no-full-session > gpurun_out/r6h_profile.log 2>&1
tail -3 gpurun_out/r6h_profile.log | cut -c1-160
cd /tmp; rm -rf /tmp/ks1 /tmp/mu
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks1 --output-format csv -- python $R/bench.py --config 3 --lanes 1 --steps 274 --warmup 137 --no-full-session --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > $R/gpurun_out/r6h_config3_lanes1_stats_bench.json 2> /tmp/ks1.err
cp "$(find /tmp/ks1 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r6h_config3_lanes1_kernel_stats.csv
timeout 600 rocprofv3 --pmc MfmaUtil -d /tmp/mu --output-format csv -- python $R/bench.py --config 3 --lanes 1 --steps 137 --warmup 137 --no-full-session --cpu-frames 0 --exact-f32-steps 0 --profile-every 0 > /dev/null 2> /tmp/mu.err
python $R/scripts/pmc_mfma_util.py "$(find /tmp/mu -name '*counter_collection.csv' | head -1)" $R/gpurun_out/r6h_config3_mfma_util.json | head -5 | cut -c1-170
cd $R
el profiles
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6h_bench_config3_driverflags.json 2> gpurun_out/r6h_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r6h_bench_config3_driverflags.json').read().strip().splitlines()[-1])
print('bench', d['value'], 'one lane', (d.get('one_clip_in_flight') or {}).get('value'), 'full', d['full_session']['value'], d['full_session']['one_clip_in_flight']['value'], 'frac', d['roofline']['frac'], d['roofline']['timed_region']['frac'], 'parity', d['parity']['min_iou_engine_vs_ref_fp32'], d['parity']['fp64']['gate_passed'])"
el bench
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r6h_pytest.log 2>&1
echo "full pytest rc $?"; tail -3 gpurun_out/r6h_pytest.log | cut -c1-250
cp gpurun_out/parity_ratios.jsonl gpurun_out/r6h_parity_ratios.jsonl 2>/dev/null
echo "total $(( $(date +%s) - t0 )) s"

#!/bin/bash
# round 5, call 3: (1) fp16 subnormal operands in the matrix pipe (ubench) and the f16x3 convolutions' error against input magnitude, (2) new GPU tests,
# (3) two sessions in flight: launch-geometry hint (chip_share) and tile rules A/B, (4) lanes for config 2 / config 4
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo "[t=$(( $(date +%s) - t0 )) s] $*"; }
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_f16_rounding.hip -o /tmp/mfr 2>/dev/null && /tmp/mfr > gpurun_out/r6c_mfma_rounding.txt 2>&1
tail -5 gpurun_out/r6c_mfma_rounding.txt | cut -c1-250
timeout 200 python scripts/conv_small_input_error.py > gpurun_out/r6c_conv_small_input_error.jsonl 2> gpurun_out/r6c_conv_small.err
python - <<'PY'
import json
for l in open('gpurun_out/r6c_conv_small_input_error.jsonl'):
    if l.startswith('{'):
        d = json.loads(l)
        print('scale', d['scale'], ' '.join('%s rms %.2e max %.2e mean %+.1e' % (k, v['rel_rms'], v['rel_max'], v['rel_mean']) for k, v in d.items() if isinstance(v, dict)))
PY
el numerics
timeout 900 python -m pytest -q -x "tests/test_gpu_engine.py::test_gui_call_pattern_under_autocast_golden" "tests/test_gpu_engine.py::test_concurrent_passes_and_suite_lanes_are_bit_identical" tests/test_dataset.py tests/test_gpu_train.py tests/test_gpu_ops.py -m gpu > gpurun_out/r6c_pytest.log 2>&1
echo "pytest rc $?"; tail -6 gpurun_out/r6c_pytest.log | cut -c1-400
el tests
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
ab() {   # name, bench args..., then env after --
  name=$1; shift
  a=$(env "$@" timeout 200 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], (d.get('one_clip_in_flight') or {}).get('value'))")
  b=$(env "$@" timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], (d.get('one_clip_in_flight') or {}).get('value'))")
  echo "$name | driver window: $a | 2 sessions: $b | t=$(( $(date +%s) - t0 ))" | tee -a gpurun_out/r6c_ab.txt
}
rm -f gpurun_out/r6c_ab.txt
ab lanes2_share MIVOS_BENCH_LANES=2
ab lanes2_noshare MIVOS_BENCH_LANES=2 MIVOS_PP_SHARE_CAP=0
ab lanes2_share_small128 MIVOS_BENCH_LANES=2 MIVOS_PP_SMALL_WGS=128
ab lanes2_share_small256 MIVOS_BENCH_LANES=2 MIVOS_PP_SMALL_WGS=256
ab lanes2_nosplit MIVOS_BENCH_LANES=2 MIVOS_PP_SPLIT=1
ab lanes2_nosplit_small128 MIVOS_BENCH_LANES=2 MIVOS_PP_SPLIT=1 MIVOS_PP_SMALL_WGS=128
ab lanes2_share_b MIVOS_BENCH_LANES=2
ab lanes3_share MIVOS_BENCH_LANES=3
el ab
for L in 1 2 3 4; do
  timeout 200 $B --config 2 --lanes $L --steps 138 --warmup 69 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 lanes', d['config'].get('clips_in_flight_per_gpu'), d['value'], (d.get('one_clip_in_flight') or {}).get('value'))" | tee -a gpurun_out/r6c_ab.txt
done
for L in 2 3 4; do
  timeout 300 python bench.py --config 4 --clips 48 --lanes $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config4 48 clips lanes', d['config'].get('clips_in_flight_per_gpu'), d['value'], 'frames/s checksum', d['config']['suite_checksum'])" | tee -a gpurun_out/r6c_ab.txt
done
el lanes
echo "total $(( $(date +%s) - t0 )) s"

#!/bin/bash
# round 4, call 2: the tests call 1 did not finish or failed (entry script with the conditioned fixture, 1080p teacher-forced step, checkpoint round
# trip ...), the long closed-loop session against the fp32 (+ fp64) oracle computed beforehand on the builder's CPU (gpurun_in/), knob A/Bs.
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
rm -f gpurun_out/teacher_forced.jsonl gpurun_out/entry_script_parity.jsonl
timeout 900 python -m pytest -q -x tests/test_entry_script.py tests/test_dataset.py tests/test_gpu_teacher_forced.py tests/test_gpu_train.py "tests/test_gpu_engine.py::test_fp16_range_overflow_is_detected" "tests/test_gpu_engine.py::test_update_mask_only_golden" -m gpu --durations=8 > gpurun_out/r5b_pytest.log 2>&1
echo "pytest rc $? after $(( $(date +%s) - t0 )) s"; tail -30 gpurun_out/r5b_pytest.log | cut -c1-300
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
run() {   # name, env...
  name=$1; shift
  a=$(env "$@" timeout 200 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  b=$(env "$@" timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$name | driver window: $a | 2 sessions: $b | t=$(( $(date +%s) - t0 ))" | tee -a gpurun_out/r5b_ab.txt
}
rm -f gpurun_out/r5b_ab.txt
run base0 X=1
run qbatch10 MIVOS_QUERY_BATCH=10
run qbatch20 MIVOS_QUERY_BATCH=20
run small128 MIVOS_PP_SMALL_WGS=128
run wide96 MIVOS_PP_WIDE_NK=96
run base1 X=1
run small256 MIVOS_PP_SMALL_WGS=256
run thr2 MIVOS_PP_SPLIT_THR=2
echo "total $(( $(date +%s) - t0 )) s"

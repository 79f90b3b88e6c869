#!/bin/bash
# round 4, call 2: A/B of host-side / tile-selection knobs on the driver's window (--steps 20 --warmup 5) and on 2 full sessions
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0"
run() {   # name, env...
  name=$1; shift
  a=$(env "$@" timeout 200 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  b=$(env "$@" timeout 200 $B --steps 274 --warmup 137 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "$name | driver window: $a | 2 sessions: $b" | tee -a gpurun_out/r5b_ab.txt
}
rm -f gpurun_out/r5b_ab.txt
run base0 X=1
run qbatch10 MIVOS_QUERY_BATCH=10
run qbatch20 MIVOS_QUERY_BATCH=20
run base1 X=1
run small128 MIVOS_PP_SMALL_WGS=128
run small256 MIVOS_PP_SMALL_WGS=256
run wide96 MIVOS_PP_WIDE_NK=96
run base2 X=1
run thr2 MIVOS_PP_SPLIT_THR=2
run thr6 MIVOS_PP_SPLIT_THR=6
run qb20_small128 MIVOS_QUERY_BATCH=20 MIVOS_PP_SMALL_WGS=128
run base3 X=1

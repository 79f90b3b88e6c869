#!/bin/bash
# round 4, call 8: the long closed-loop session with the engine in its exact-fp32 mode (same arithmetic class as the reference): how much of the
# engine-vs-fp64 distance of the default precision is the f16x3 operand format
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python scripts/long_session_parity.py engine --precision f32 --ref32 gpurun_in/long32 --ref64 gpurun_in/long64 --wait 5 --json gpurun_out/r5h_long_session_parity_exact_f32.json > gpurun_out/r5h_long_engine.log 2>&1
tail -3 gpurun_out/r5h_long_engine.log | cut -c1-1500

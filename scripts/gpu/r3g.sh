#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_engine.py -q -k "fusion_net or fusion_generator or e2e_small or small_session" > $O/r3g_tests.log 2>&1; tail -6 $O/r3g_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

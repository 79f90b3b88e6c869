#!/bin/bash
# round 6, final commit: the driver's GPU check (pytest -x -m gpu, smoke) for the record
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
rm -f gpurun_out/parity_ratios.jsonl gpurun_out/teacher_forced.jsonl gpurun_out/long_horizon_parity.jsonl gpurun_out/entry_script_parity.jsonl
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r7s_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r7s_pytest.log | cut -c1-250
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

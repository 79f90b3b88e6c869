#!/bin/bash
# round 5, last call: the driver's GPU check on the final tree
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 640 python -m pytest tests -x -q -m gpu > gpurun_out/r6n_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r6n_pytest.log | cut -c1-200

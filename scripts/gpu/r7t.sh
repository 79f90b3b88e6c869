#!/bin/bash
set +e
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "chip_share" 2>&1 | tail -4 | cut -c1-300

#!/bin/bash
# round-2 GPU call M: S2M / clip_io tests, dilated conv cases, full suite timing, config 5 memread variant check
set +e
export TMPDIR=/tmp
O=gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_s2m.py tests/test_clip_io.py -m gpu -q -s > $O/r2m_new.log 2>&1; grep -E "S2M|to_mask|passed|failed|Error|error|assert" $O/r2m_new.log | tail -25
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/r2m_pytest.log 2>&1; tail -25 $O/r2m_pytest.log
echo "== memread microbench"; timeout 300 python scripts/memread_microbench.py > $O/r2m_memread.txt 2>&1; tail -9 $O/r2m_memread.txt

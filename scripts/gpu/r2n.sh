#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out
echo "== s2m + clip io + 1080p + headline K=2"; timeout 900 python -m pytest tests/test_gpu_s2m.py tests/test_clip_io.py tests/test_gpu_engine.py -m gpu -q -s -k "s2m or davis or ingest or 1080p_three or 2-50-5" > $O/r2n_new.log 2>&1; grep -E "S2M|to_mask|1080p|K=2|passed|failed|Error|error|assert" $O/r2n_new.log | tail -25

#!/bin/bash
# round 3, call 7: any-k read, batched S2M, overflow guards, s2m bench
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_train.py tests/test_gpu_s2m.py -m gpu -q -rP -k "large_k or s2m or davis_processor or overflow or adam or topk_out_of_range or topk_larger or memory_read_golden or callbacks" > gpurun_out/r4g_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r4g_pytest.log; grep -E "^S2M|^to_mask|^E  " gpurun_out/r4g_pytest.log | cut -c1-250 | head -20
timeout 200 python bench.py --config s2m 2> gpurun_out/r4g_s2m.err | tee gpurun_out/r4g_bench_s2m.json | cut -c1-400
timeout 200 python bench.py --config s2m --objects 1 2>/dev/null | cut -c1-200
timeout 300 python bench.py --config 4 --clips 16 2> gpurun_out/r4g_c4.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['value'], d['cost_model'])"

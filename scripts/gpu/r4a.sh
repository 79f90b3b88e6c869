#!/bin/bash
# round 3, call 1: full GPU test-suite (new deep-bank / fp64-gated tests) + the driver's bench command
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity_ratios.jsonl
timeout 1300 python -m pytest tests -m gpu -q -rP --durations=12 > gpurun_out/r4a_pytest.log 2>&1
echo "pytest rc $?"
tail -5 gpurun_out/r4a_pytest.log
grep -E "^deep bank|worst per-frame ratio" gpurun_out/r4a_pytest.log | cut -c1-400
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench_driverflags.json 2> gpurun_out/r4a_bench_driverflags.err
echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r4a_bench_driverflags.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['full_session'], d['parity'], d['cpu_baseline']['seconds'])
PY

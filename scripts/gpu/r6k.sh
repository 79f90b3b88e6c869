#!/bin/bash
# round 5, call 11: the auxiliary bench modes on the final tree (training step, S2M, generator suite)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for a in "--config train" "--config s2m" "--config 4 --generator --clips 8"; do
  timeout 400 python bench.py $a 2> gpurun_out/r6k_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a ->', d['metric'][:60], d['value'], d['unit'], d.get('ms_per_step'))" || tail -5 gpurun_out/r6k_err.txt
done

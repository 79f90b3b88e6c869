#!/bin/bash
# round 3: the 256-query select kernel as the default from 400 k positions: config 5 line, kernel stats, PMC traffic of one deep read
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 260 python bench.py --config 5 2>/dev/null | tail -1 > gpurun_out/r4m_bench_config5.json
python -c "
import json
d=json.loads(open('gpurun_out/r4m_bench_config5.json').read()); r=d['roofline']; print('config5', d['value'], d['ms_per_step'], r['affinity'], d.get('parity'))" | cut -c1-900
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc_$c
  timeout 90 rocprofv3 --pmc $c -d /tmp/pc_$c --output-format csv -- python $R/scripts/memread_case.py 3 100 8160 50 q256 > /dev/null 2> /tmp/pc_$c.err
  echo "case pmc $c rc $?"
done
python $R/scripts/pmc_traffic.py $(find /tmp/pc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $R/gpurun_out/r4m_config5_memread256_T100_pmc_traffic.json | head -6
rm -rf /tmp/kt
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/kt --output-format csv -- python $R/bench.py --config 5 --frames 262 --cpu-frames 0 --exact-f32-steps 0 > /dev/null 2> /tmp/kt.err
echo "kernel trace rc $?"
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r4m_config5_kernel_stats.csv && head -8 $f | cut -c1-160

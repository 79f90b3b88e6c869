#!/bin/bash
# cycle breakdown of the 480p select kernel after count_kth (workgroup 0 / wave 0)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in "5 7 1620 50" "5 5 1620 50" "1 5 1620 20" "5 12 1620 50"; do
MIVOS_MEMREAD_DBG=1 timeout 60 python scripts/memread_case.py $c q64 2>&1 | grep "memread_select" | tail -1
done | tee gpurun_out/r4o_memread_cycles.txt

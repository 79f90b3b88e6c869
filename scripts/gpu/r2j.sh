#!/bin/bash
# round-2 GPU call J: SALU-free selection slices
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== memread tests"; timeout 300 python -m pytest tests/test_gpu_ops.py -k memory_read -q > $O/r2j_memtest.log 2>&1; rc1=$?; tail -4 $O/r2j_memtest.log
echo "== memread microbench"; timeout 300 python scripts/memread_microbench.py --check > $O/r2j_memread.txt 2>&1; tail -11 $O/r2j_memread.txt
echo "== cycles per tile"; MIVOS_MEMREAD_DBG=1 timeout 200 python scripts/memread_microbench.py 2>&1 | grep -E "memread_select\]" | awk 'NR%6==0' > $O/r2j_memread_cycles.txt; cat $O/r2j_memread_cycles.txt
if [ $rc1 -ne 0 ]; then echo "memread tests fail: stopping"; exit 0; fi
echo "== engine tests (fast subset)"; timeout 600 python -m pytest tests/test_gpu_engine.py -q -k "not headline and not 1080p" > $O/r2j_pytest.log 2>&1; tail -5 $O/r2j_pytest.log
echo "== bench config 3 (2 sessions)"; timeout 600 python bench.py --steps 274 --cpu-frames 0 --exact-f32-steps 0 > $O/r2j_bench_c3.json 2> $O/r2j_bench_c3.err; cut -c1-200 $O/r2j_bench_c3.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2j_bench_c3.json")); a=d["roofline"]["affinity"]; print("affinity in situ:", a["achieved"], a["frac"], a["avg_launch_us"])
PY

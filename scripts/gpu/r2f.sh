#!/bin/bash
# round-2 GPU call F: decoder restructure (v16 partial sums, SH32 producers, Cout=1 projection): tests + bench with per-shape table
set +e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== pytest gpu (without the two fp64-arbitration sessions)"; timeout 900 python -m pytest tests -m gpu -q -k "not headline" --durations=8 > $O/r2f_pytest.log 2>&1; tail -30 $O/r2f_pytest.log
echo "== bench config 3 (2 sessions, per-shape table)"; MIVOS_BENCH_SHAPES=1 timeout 600 python bench.py --steps 274 --cpu-frames 0 --exact-f32-steps 0 > $O/r2f_bench_c3.json 2> $O/r2f_bench_c3.err; cut -c1-300 $O/r2f_bench_c3.json; grep "^#" $O/r2f_bench_c3.err | head -60
echo "== bench config 2"; timeout 300 python bench.py --config 2 --cpu-frames 0 > $O/r2f_bench_c2.json 2> $O/r2f_bench_c2.err; cut -c1-200 $O/r2f_bench_c2.json

#!/bin/bash
# round 4, call 10: upsample2x_add_multi with the objects of a pixel adjacent in the walk (broadcast skip read once): tests, kernel time, traffic
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$(pwd)
timeout 200 python -m pytest -q tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -k "upsample or decoder_golden or end_to_end or segment_with_query or single_step or teacher" 2>&1 | tail -3
cd /tmp
ARGS="--cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 --steps 137 --warmup 8"
rm -rf /tmp/ks /tmp/pm_F
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ks --output-format csv -- python $R/bench.py $ARGS > /tmp/ks.json 2> /tmp/ks.err
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r5i_config3_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/r5i_config3_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("upsample", "128, 256", "maxpool")): print(r["Name"][:50], r["Calls"], round(float(r["AverageNs"])/1e3, 1), "us")
PY
timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/pm_F --output-format csv -- python $R/bench.py $ARGS > /dev/null 2> /tmp/pm_F.err
python - <<PY
import csv, collections, glob
f = glob.glob("/tmp/pm_F/**/*counter_collection.csv", recursive=True)[0]
a = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "upsample" in r["Kernel_Name"]:
        a["upsample"][0] += float(r["Counter_Value"]); a["upsample"][1] += 1
for k, (v, n) in a.items(): print(k, "read MB/launch", round(2 * v * 1024 / n / 1e6, 1), "launches", n)
PY
cd $R
python bench.py --steps 20 --warmup 5 --cpu-frames 0 --exact-f32-steps 0 --no-full-session --profile-every 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver window', d['value'], d['ms_per_step'])"

#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; do
  rm -rf /tmp/pm_$c
  timeout 300 rocprofv3 --pmc $c -d /tmp/pm_$c --output-format csv -- python $R/scripts/memread_case.py 3 100 8160 50 q64 q128 > /dev/null 2> /tmp/pm_$c.err
  f=$(find /tmp/pm_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, set()])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    a = agg[r["Kernel_Name"].split("(")[0][-60:]]; a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
for k, (v, d) in agg.items():
    if "select" in k: print(sys.argv[2], k, "per launch:", v / len(d), "launches", len(d))
PY
done 2>&1 | tee $O/r2y_pmc_memread_1080p.txt

"""Sum rocprofv3 --pmc counter_collection.csv per kernel (averaged per dispatch)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    n = len(disp[k])
    print(k, "dispatches", n)
    for a, b in sorted(v.items()):
        print(f"   {a:32s} {b / n:16.0f}")

"""Per-kernel MFMA utilisation from a `rocprofv3 --pmc MfmaUtil` counter_collection.csv (derived counter of rocprofiler-sdk:
sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) x SIMD_NUM) x 100, i.e. the share of matrix-pipe cycles that were busy
while the dispatch ran): mean, min, max over the dispatches of every kernel, most dispatches first.  `out.json` optional."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_fingerprint import csrc_fingerprint  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "MfmaUtil":
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
table = {}
print("kernel | dispatches | MfmaUtil mean % (min - max)")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if max(v) <= 0.0:
        continue
    table[k] = dict(dispatches=len(v), mfma_util_mean_pct=round(sum(v) / len(v), 2), min_pct=round(min(v), 2), max_pct=round(max(v), 2))
    print(f"{k[:110]:110s} {len(v):6d} {sum(v) / len(v):7.2f} ({min(v):6.2f} - {max(v):6.2f})")
if len(sys.argv) > 2:
    table["_meta"] = dict(csrc_fingerprint=csrc_fingerprint(), counter="MfmaUtil (rocprofv3 derived counter)")
    json.dump(table, open(sys.argv[2], "w"), indent=1)

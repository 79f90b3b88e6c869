#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats summary (…kernel_stats.csv) -> a markdown table with time per propagated frame.
    python scripts/kernel_stats_md.py kernel_stats.csv <frames in the process> "<title>" > profiles/<round>_kernel_stats.md"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2])
total = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print(f"# {sys.argv[3]}\n\ntotal kernel time {total:.1f} ms = {total / frames:.3f} ms per propagated frame\n")
print("| kernel | calls | avg us | total ms | ms / frame | % |\n|---|---|---|---|---|---|")
for r in rows[:26]:
    name = r["Name"].replace("void ", "").replace("mivos::", "").split("(")[0]
    print(f"| `{name[:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.1f} | {float(r['TotalDurationNs']) / 1e6 / frames:.3f} | {float(r['Percentage']):.1f} |")

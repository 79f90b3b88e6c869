"""Build profiles/<round>_pmc_traffic.json from two separate rocprofv3 --pmc passes of bench.py.

    rocprofv3 --pmc FETCH_SIZE -d /tmp/pm_FETCH_SIZE --output-format csv -- python bench.py ... --cpu-frames 0 --profile-every 0
    rocprofv3 --pmc WRITE_SIZE -d /tmp/pm_WRITE_SIZE --output-format csv -- python bench.py ... --cpu-frames 0 --profile-every 0
    python scripts/pmc_traffic.py FETCH.csv WRITE.csv profiles/r01g_pmc_traffic.json

FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md §HBM: the counter reports 64 B per 128 B request);
both counters are in KiB.  Values are averaged per dispatch of each kernel."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_fingerprint import csrc_fingerprint  # noqa: E402


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1].add(r["Dispatch_Id"])
    return {k: (v / len(d), len(d)) for k, (v, d) in agg.items()}


def main(fetch_csv, write_csv, out_json):
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    rows = [(k, fetch[k][1], fetch[k][0], write[k][0]) for k in fetch if k in write]
    rows.sort(key=lambda r: -(2 * r[2] + r[3]) * r[1])
    out = {}
    print("kernel | launches | read MB/launch (FETCH_SIZE x2) | write MB/launch")
    for k, n, f, w in rows[:16]:
        short = k.split("(")[0].replace("void ", "")
        print(f"{short[:72]:72s} {n:5d} {2 * f * 1024 / 1e6:10.1f} {w * 1024 / 1e6:10.1f}")
        out[short] = dict(launches=n, read_bytes_per_launch=2 * f * 1024, write_bytes_per_launch=w * 1024,
                          fetch_size_raw_kb=f, write_size_raw_kb=w)
    out["_meta"] = dict(csrc_fingerprint=csrc_fingerprint(), counters="FETCH_SIZE x 2 (gfx950 correction) and WRITE_SIZE, separate rocprofv3 --pmc passes, KiB -> bytes")
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])

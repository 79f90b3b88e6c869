#!/usr/bin/env python
"""Per-layer-shape IN-SITU kernel times of the LDS-DMA convolutions: joins a rocprofv3 kernel trace of a bench run with the library's launch log.

    MIVOS_CONV_LOG=/tmp/conv.log rocprofv3 --kernel-trace -d /tmp/kt --output-format csv -- python bench.py --lanes 1 ...
    python scripts/insitu_shape_table.py /tmp/kt /tmp/conv.log [--json out.json] [--label default]
    python scripts/insitu_shape_table.py --diff a.json b.json          # A/B of two rule sets (same box)

The library writes one line per `conv_f16x3_pp_kernel` launch in host order (csrc/conv_f16x3_dma.hip, MIVOS_CONV_LOG); rocprofv3's Dispatch_Id
follows the same order (one host thread issues every launch), so the n-th pp dispatch of the trace is the n-th line of the log - checked launch by
launch against the grid size.  Per shape: launches, average kernel duration, average duration of the split-K reduce that follows it, and the
average gap between the end of the previous kernel on the same HSA queue (= HIP stream) and this kernel's start."""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find_trace(d):
    if os.path.isfile(d):
        return d
    hits = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not hits:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    return max(hits, key=os.path.getsize)


def load_trace(path):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append(dict(id=int(r["Dispatch_Id"]), queue=r.get("Queue_Id", "0"), name=r["Kernel_Name"], t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"]),
                             gx=int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), gy=int(r.get("Grid_Size_Y", 1) or 1), wx=int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    rows.sort(key=lambda r: r["id"])
    return rows


def table(trace_dir, log_path, skip_frames_before=0.0):
    rows = load_trace(find_trace(trace_dir))
    log = [ln.split() for ln in open(log_path) if ln.startswith("pp ")]
    by_queue_prev_end = {}
    shapes = defaultdict(lambda: dict(n=0, us=0.0, reduce_us=0.0, reduces=0, gap_us=0.0, gaps=0, flop=0.0))
    li = 0
    mismatches = 0
    pending_reduce = None
    t_first, t_last = rows[0]["t0"], max(r["t1"] for r in rows)
    busy = defaultdict(float)
    for r in rows:
        prev_end = by_queue_prev_end.get(r["queue"])
        by_queue_prev_end[r["queue"]] = r["t1"]
        busy[r["queue"]] += (r["t1"] - r["t0"]) * 1e-3
        if "conv_f16x3_pp_kernel" in r["name"]:
            if li >= len(log):
                mismatches += 1
                continue
            _, bm, bn, M, cin, cout, k, stride, res, slices, wgs, share, yfmt = log[li]
            li += 1
            folded = int(slices) < 0                 # (a negative slice count: the slices ran folded inside one workgroup per tile - gridDim.y == 1, no reduce launch)
            if int(wgs) * 512 != r["gx"] or (1 if folded else int(slices)) != r["gy"]:
                mismatches += 1
            key = (f"{bm}x{bn}" + ("f" if folded else ""), int(M), int(cin), int(cout), int(k), int(stride), bool(int(res)), abs(int(slices)))
            s = shapes[key]
            s["n"] += 1
            s["us"] += (r["t1"] - r["t0"]) * 1e-3
            s["flop"] += 2.0 * int(M) * int(cout) * int(k) * int(k) * int(cin)
            if prev_end is not None:
                s["gap_us"] += max(0.0, (r["t0"] - prev_end) * 1e-3)
                s["gaps"] += 1
            pending_reduce = key if int(slices) > 1 else None      # (folded launches have no reduce pass)
        elif "splitk_reduce_kernel" in r["name"] and pending_reduce is not None:
            s = shapes[pending_reduce]
            s["reduce_us"] += (r["t1"] - r["t0"]) * 1e-3
            s["reduces"] += 1
            pending_reduce = None
    out = []
    for key, s in shapes.items():
        tot = s["us"] + s["reduce_us"]
        out.append(dict(tile=key[0], M=key[1], Cin=key[2], Cout=key[3], k=key[4], stride=key[5], res=key[6], slices=key[7], n=s["n"],
                        avg_us=round(s["us"] / s["n"], 2), avg_reduce_us=round(s["reduce_us"] / max(s["reduces"], 1), 2), avg_total_us=round(tot / s["n"], 2),
                        avg_gap_before_us=round(s["gap_us"] / max(s["gaps"], 1), 2), tflops=round(s["flop"] / tot / 1e6, 1), total_ms=round(tot * 1e-3, 3)))
    out.sort(key=lambda r: -r["total_ms"])
    meta = dict(trace=os.path.basename(find_trace(trace_dir)), log_lines=len(log), pp_dispatches=li, order_mismatches=mismatches, wall_ms=round((t_last - t_first) * 1e-6, 2),
                queue_busy_ms={q: round(v * 1e-3, 2) for q, v in busy.items()}, kernel_ms=round(sum(busy.values()) * 1e-3, 2))
    return dict(meta=meta, shapes=out)


def fmt(t):
    lines = ["# " + json.dumps(t["meta"]), f"{'tile':8s} {'M':>7s} {'Cin':>5s} {'Cout':>5s} k s res sl {'n':>5s} {'avg us':>8s} {'+reduce':>8s} {'gap':>6s} {'TF/s':>6s} {'total ms':>9s}"]
    for r in t["shapes"]:
        lines.append(f"{r['tile']:8s} {r['M']:7d} {r['Cin']:5d} {r['Cout']:5d} {r['k']} {r['stride']} {int(r['res'])}   {r['slices']} {r['n']:5d} {r['avg_us']:8.1f} {r['avg_reduce_us']:8.1f} "
                     f"{r['avg_gap_before_us']:6.1f} {r['tflops']:6.1f} {r['total_ms']:9.2f}")
    return "\n".join(lines)


def diff(a, b):
    ka = {(r["M"], r["Cin"], r["Cout"], r["k"], r["stride"], r["res"]): r for r in a["shapes"]}
    kb = {(r["M"], r["Cin"], r["Cout"], r["k"], r["stride"], r["res"]): r for r in b["shapes"]}
    lines = [f"# A: {json.dumps(a['meta'])}", f"# B: {json.dumps(b['meta'])}",
             f"{'M':>7s} {'Cin':>5s} {'Cout':>5s} k s res | {'A tile/sl':>10s} {'A us':>8s} | {'B tile/sl':>10s} {'B us':>8s} | {'B/A':>6s} {'n':>5s} {'saved ms':>9s}"]
    tot = 0.0
    for key in sorted(set(ka) & set(kb), key=lambda k: -(ka[k]["total_ms"])):
        x, y = ka[key], kb[key]
        if x["tile"] == y["tile"] and x["slices"] == y["slices"] and abs(x["avg_total_us"] - y["avg_total_us"]) < 0.03 * x["avg_total_us"]:
            continue
        saved = (x["avg_total_us"] - y["avg_total_us"]) * min(x["n"], y["n"]) * 1e-3
        tot += saved
        lines.append(f"{key[0]:7d} {key[1]:5d} {key[2]:5d} {key[3]} {key[4]} {int(key[5])}   | {x['tile'] + '/' + str(x['slices']):>10s} {x['avg_total_us']:8.1f} | "
                     f"{y['tile'] + '/' + str(y['slices']):>10s} {y['avg_total_us']:8.1f} | {y['avg_total_us'] / x['avg_total_us']:6.3f} {x['n']:5d} {saved:9.2f}")
    lines.append(f"# conv time saved by B over the traced region: {tot:.2f} ms (wall A {a['meta']['wall_ms']} ms, B {b['meta']['wall_ms']} ms; kernel time A {a['meta']['kernel_ms']}, B {b['meta']['kernel_ms']})")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="*")
    ap.add_argument("--json", default=None)
    ap.add_argument("--label", default=None)
    ap.add_argument("--diff", action="store_true")
    args = ap.parse_args()
    if args.diff:
        a, b = (json.load(open(p)) for p in args.paths[:2])
        print(diff(a, b))
        return
    t = table(args.paths[0], args.paths[1])
    t["meta"]["label"] = args.label
    if args.json:
        with open(args.json, "w") as f:
            json.dump(t, f)
    print(fmt(t))


if __name__ == "__main__":
    main()

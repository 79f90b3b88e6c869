#!/usr/bin/env python
"""Timeline facts of a bench run from its rocprofv3 kernel trace (the raw *kernel_trace.csv, optionally gzipped): queue occupancy, launch-to-launch gaps
on the main queue, per-frame time of plain / fused frames (median and mean: every QUERY_BATCH-th plain frame carries the next batch's encoding), kernel time
per plain / fused frame on the main queue, and every GPU-idle gap (no queue busy) above a threshold with the kernels around it.

    python scripts/trace_timeline.py gpurun_out/r7a_kernel_trace_default.csv.gz --session 137 --plain 69 > profiles/r06a_timeline.md
"""
import argparse
import collections
import csv
import gzip

import numpy as np


def load(path):
    op = gzip.open if path.endswith(".gz") else open
    rows = []
    with op(path, "rt") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mivos::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"])))
    rows.sort()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--session", type=int, default=137, help="steps per session")
    ap.add_argument("--plain", type=int, default=69, help="plain steps at the start of a session")
    ap.add_argument("--skip-sessions", type=int, default=1, help="sessions at the start of the trace that are warm-up")
    ap.add_argument("--gap-ms", type=float, default=1.0)
    a = ap.parse_args()
    rows = load(a.trace)
    sel = [r for r in rows if "memread_select" in r[3]]
    mainq = sel[0][2]
    st = np.array([r[0] for r in sel])
    print(f"# timeline of {a.trace}\n")
    print(f"{len(rows)} kernel dispatches, {len(sel)} propagated frames (one select launch each), main queue = {mainq}\n")
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    print("| queue | dispatches | busy ms | launch-to-launch gaps < 20 us: count / total ms / mean us |\n|---|---|---|---|")
    for q, rs in sorted(byq.items()):
        gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
        small = [g for g in gaps if 0 <= g < 20000]
        print(f"| {q} | {len(rs)} | {sum(r[1] - r[0] for r in rs) / 1e6:.1f} | {len(small)} / {sum(small) / 1e6:.2f} / {sum(small) / max(len(small), 1) / 1e3:.2f} |")
    d = np.diff(st) / 1e6
    print("\n| session | plain frames: median / mean ms | fused frames: median / mean ms |\n|---|---|---|")
    n_sess = len(sel) // a.session
    for s in range(n_sess):
        p = d[s * a.session:s * a.session + a.plain]
        f = d[s * a.session + a.plain:(s + 1) * a.session - 1]
        if len(p) > 2 and len(f) > 2:
            print(f"| {s} | {np.median(p):.2f} / {p[1:-1].mean():.2f} | {np.median(f):.2f} / {f[1:-1].mean():.2f} |")

    def frame_stats(lo, hi):
        agg = collections.defaultdict(lambda: [0, 0])
        for r in rows:
            if st[lo] <= r[0] < st[hi] and r[2] == mainq:
                key = r[3] + (f" g{r[4]}x{r[5]}" if "pp_kernel" in r[3] else "")
                agg[key][0] += 1
                agg[key][1] += r[1] - r[0]
        n = hi - lo
        return {k: (v[0] / n, v[1] / 1e3 / v[0], v[1] / 1e6 / n) for k, v in agg.items()}
    s0 = a.skip_sessions * a.session
    if len(sel) >= s0 + a.session:
        p = frame_stats(s0 + 10, s0 + a.plain - 9)
        f = frame_stats(s0 + a.plain + 5, s0 + a.session - 8)
        keys = sorted(set(p) | set(f), key=lambda k: -(p.get(k, (0, 0, 0))[2] + f.get(k, (0, 0, 0))[2]))
        print("\nMain-queue kernels per frame (g<workgroups>x<K slices> for the LDS-DMA convolutions):\n")
        print("| kernel | plain: launches/frame, avg us, ms/frame | fused: launches/frame, avg us, ms/frame |\n|---|---|---|")
        tp = tf = 0.0
        for k in keys[:30]:
            x, y = p.get(k, (0, 0, 0)), f.get(k, (0, 0, 0))
            tp += x[2]; tf += y[2]
            print(f"| `{k[:60]}` | {x[0]:.1f}, {x[1]:.1f}, {x[2]:.3f} | {y[0]:.1f}, {y[1]:.1f}, {y[2]:.3f} |")
        print(f"| (sum of the rows) | {tp:.2f} | {tf:.2f} |")
    t_lo = st[s0] if len(st) > s0 else rows[0][0]
    allr = [r for r in rows if r[0] >= t_lo]
    print(f"\nGPU-idle gaps >= {a.gap_ms} ms (no queue busy) after the warm-up session(s):\n")
    print("| gap ms | at ms | last kernel before | first kernel after |\n|---|---|---|---|")
    cur_end, last, total = allr[0][1], allr[0], 0.0
    for r in allr[1:]:
        if r[0] > cur_end + a.gap_ms * 1e6:
            print(f"| {(r[0] - cur_end) / 1e6:.2f} | {(cur_end - t_lo) / 1e6:.1f} | q{last[2]} `{last[3][:40]}` | q{r[2]} `{r[3][:40]}` |")
            total += (r[0] - cur_end) / 1e6
        if r[1] > cur_end:
            cur_end, last = r[1], r
    print(f"\ntotal {total:.1f} ms of {(cur_end - t_lo) / 1e6:.0f} ms")


if __name__ == "__main__":
    main()

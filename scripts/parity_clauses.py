#!/usr/bin/env python
"""List which closed-loop parity records (tests/test_gpu_engine.py::fp64_gate -> gpurun_out/parity_ratios.jsonl) did not pass
through the strict clause, frame by frame, with the quantile in force.

    python scripts/parity_clauses.py profiles/r04a_parity_ratios.jsonl
"""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
print(f"{len(rows)} gate records; strict clause: |engine - fp64| <= 1.5 |reference_fp32 - fp64| + 2.5e-4 per frame (factor 2 until round 4)")
tie = 0
for r in rows:
    cl = r.get("clauses") or []
    bad = [(t, c) for t, c in enumerate(cl) if c != "strict"]
    line = (f"{r['test']:55s} frames {r['frames']:3d}  max e {r['engine_vs_fp64_max']:.2e}  max r {r['ref32_vs_fp64_max']:.2e}  worst ratio "
            f"{r['worst_frame_ratio']:6.2f}  " + ("all strict" if not bad else f"NOT strict: {bad} (tie quantile {r.get('tie_quantile')})"))
    print(line)
    for t, c in bad:
        tie += c == "tie"
        print(f"      frame {t}: e {r['per_frame_engine'][t]:.3e} r {r['per_frame_ref32'][t]:.3e}  engine quantiles {r['per_frame_engine_quantiles'][t]}  "
              f"reference quantiles {r['per_frame_ref32_quantiles'][t]}  min fp64 top-k margin of the session {r.get('min_topk_margin_fp64')}")
print(f"frames through the tie clause: {tie}; failed records: {sum(1 for r in rows if r.get('passed') is False)}")

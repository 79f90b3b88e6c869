#!/usr/bin/env python
"""Per-shape sweep of the LDS-DMA convolution kernel's tile / split-K choices on the `<128,128>` population of config 3.

    python scripts/conv_shape_sweep.py            # parent: one subprocess per (tile, split) combination (the knobs are read once per process)
    python scripts/conv_shape_sweep.py --child    # child: times every shape under the MIVOS_PP_TILE / MIVOS_PP_SPLIT of its environment

Shapes = the M = 8100 / 32 400 / 129 600 layers of the mask encoder, KeyValue and decoder that run on 128x128 tiles today (profiles/r03h_config3_conv_shapes_no_side_stream.txt).
Each launch is timed back to back on L2-warm operands (20 repetitions, HIP events), so the numbers rank the choices per shape; the in-situ effect is
then checked with `MIVOS_PP_*` on bench.py.  Output: one line per shape with the microseconds of every combination and the current default's."""
import json
import os
import subprocess
import sys

SHAPES = [  # name, N, H, W, Cin, Cout, k, stride, residual
    ("l3 1x1 1024->256", 5, 30, 54, 1024, 256, 1, 1, False), ("l3 3x3 256->256", 5, 30, 54, 256, 256, 3, 1, False),
    ("l3 1x1 256->1024 +res", 5, 30, 54, 256, 1024, 1, 1, True), ("dec 3x3 512->512", 5, 30, 54, 512, 512, 3, 1, True),
    ("kv 3x3 1024->640", 5, 30, 54, 1024, 640, 3, 1, False), ("l2 1x1 512->128", 5, 60, 108, 512, 128, 1, 1, False),
    ("l2 3x3 128->128", 5, 60, 108, 128, 128, 3, 1, False), ("l2 1x1 128->512 +res", 5, 60, 108, 128, 512, 1, 1, True),
    ("l1 1x1 64->256 +res", 5, 120, 216, 64, 256, 1, 1, True), ("l1 1x1 256->128", 5, 120, 216, 256, 128, 1, 1, False),
]
COMBOS = [(0, 0)] + [(t, s) for t in (20, 26, 21, 22, 24) for s in (1, 2, 4)]      # (MIVOS_PP_TILE, MIVOS_PP_SPLIT); (0, 0) = the library's own choice; 26 = tile 20 with MIVOS_PP_MERGE=1


def child():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mivos_amd import ops
    from mivos_amd.ops import ConvLayer
    torch.set_grad_enabled(False)
    dev, out = "cuda:0", {}
    for name, n, h, w, cin, cout, k, s, has_res in SHAPES:
        torch.manual_seed(0)
        x = ops.to_act(torch.randn(n, h, w, cin, device=dev))
        L = ConvLayer.pack(torch.randn(cout, cin, k, k) * 0.03, torch.randn(cout) * 0.1, None, s, k // 2).to(dev)
        res = ops.to_act(torch.randn(n, h, w, cout, device=dev)) if has_res else None
        y = ops.alloc_act(n, h, w, cout, x.device)
        fn = lambda: ops.conv(x, L, relu_out=True, res=res, out=y, out_act=True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps(out))


def main():
    if "--child" in sys.argv:
        return child()
    table = {}
    for tile, split in COMBOS:
        env = dict(os.environ)
        if tile:
            env["MIVOS_PP_TILE"], env["MIVOS_PP_SPLIT"] = str(20 if tile == 26 else tile), str(split)
            if tile == 26:
                env["MIVOS_PP_MERGE"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        table[(tile, split)] = json.loads(line[-1]) if line else {}
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump({f"{t}/{s}": v for (t, s), v in table.items()}, f)
    print("shape | default | " + " | ".join(f"t{t}/s{s}" for t, s in COMBOS[1:]))
    for name, *_ in SHAPES:
        print(f"{name:24s} {table[(0, 0)].get(name, '-'):>8} | " + " | ".join(f"{table[c].get(name, '-'):>7}" for c in COMBOS[1:]))


if __name__ == "__main__":
    main()

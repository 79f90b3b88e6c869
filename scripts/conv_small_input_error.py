#!/usr/bin/env python
"""Error of the f16x3 convolutions against an fp64 convolution as a function of the INPUT MAGNITUDE (GPU): the lo halves of the operand
split x = hi + lo are fp16 subnormals for |x| < 2^-3; if anything on the way treats them worse than normal numbers, the error of a layer
with small activations grows.  Prints relative rms / max errors of the LDS-DMA kernel (SH32 input), the register-staged f16x3 kernel
(fp32 input) and the exact-fp32 kernel for inputs relu(randn) * scale.

    python scripts/conv_small_input_error.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from mivos_amd import ops
    from mivos_amd.ops import ConvLayer
    torch.set_grad_enabled(False)
    dev = "cuda:0"
    n, h, w, cin, cout, k = 2, 30, 54, 256, 256, 3
    out = []
    for scale in (1.0, 0.1, 0.01, 0.001):
        torch.manual_seed(0)
        x = torch.relu(torch.randn(n, h, w, cin)) * scale
        wt = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1)
        ref32 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1).double()
        L = ConvLayer.pack(wt, None, None, 1, 1).to(dev)
        xd = x.to(dev)
        rec = dict(scale=scale, out_rms=float(ref.pow(2).mean().sqrt()))
        ops.CONV_PRECISION = "f16x3"
        y_dma = ops.to_f32(ops.conv(ops.to_act(xd), L, out_act=True)).cpu().double()
        y_reg = ops.conv(xd, L).cpu().double()
        ops.CONV_PRECISION = "f32"
        y_f32 = ops.conv(xd, L).cpu().double()
        ops.CONV_PRECISION = "f16x3"
        s = rec["out_rms"]
        for name, y in (("lds_dma_f16x3", y_dma), ("register_staged_f16x3", y_reg), ("exact_f32_mfma", y_f32), ("torch_cpu_fp32", ref32)):
            e = y - ref
            rec[name] = dict(rel_rms=float(e.pow(2).mean().sqrt() / s), rel_max=float(e.abs().max() / s), rel_mean=float(e.mean() / s))
        out.append(rec)
        print(json.dumps(rec))


if __name__ == "__main__":
    main()

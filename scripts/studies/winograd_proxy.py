#!/usr/bin/env python
"""Winograd F(2x2, 3x3) for the decoder's largest 3x3 layer - a MEASURED bound built from the engine's own kernels (VERDICT r3
item 5), before anybody writes the transform kernels.

Layer: up_8_4 out_conv, 256 -> 256, 3x3, 5 objects x 120 x 216 = 129 600 output pixels (SURVEY App. A, 152.9 GMAC for the pair).
Winograd replaces the 9-tap GEMM [129 600 x 2304] x [2304 x 256] by 16 independent GEMMs [32 400 x 256] x [256 x 256] (one per
transform-domain point; 2.25x fewer multiply-adds) plus an input transform (each 4x4 patch -> 16 values: the transformed tensor V is
4x the activation) and an output transform (16 -> 4 values per tile).

What this script times on the GPU (all with the engine's kernels, SH32 operands):
  direct    the layer as it runs today (conv_f16x3_pp_kernel<128,256>), SH32 in -> SH32 out
  gemm16    the 16 transform-domain GEMMs as ONE launch of the same kernel on a [16 x 32 400, 256] x [256, 256] problem
            (M = 518 400, K = 256, a 1x1 convolution; per-point weights would not change its shape or time), fp32 out
  v_pass    one pass that reads 133 MB of SH32 activations and writes the 531 MB transformed tensor: measured with the SH32
            pack kernel on a tensor of V's size (read fp32 531 MB + write SH32 531 MB - an UNDER-estimate of the read side is not
            possible: V has to be written once and read once by the GEMMs whatever the fusion)
  m_pass    the output transform's traffic: read 531 MB of fp32 products, write 133 MB: measured with the SH32 pack kernel on
            a quarter-size output (read 531 MB / write 531 MB upper bound is v_pass; lower bound = 664 MB at the measured rate)
Winograd time >= gemm16 + (V written by the producer: +398 MB of extra writes) + m_pass-like read of the products, unless the output
transform is fused into the GEMM (which needs all 16 points of a tile in one workgroup: 16 accumulator sets, 32 x 64 tiles per
point - a quarter of the arithmetic intensity per LDS byte of the 128 x 256 tile the direct kernel uses)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mivos_amd import ops  # noqa: E402
from mivos_amd.ops import ConvLayer  # noqa: E402

torch.set_grad_enabled(False)
DEV = "cuda:0"
REPS = 20


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3      # us


def main():
    torch.manual_seed(0)
    n, h, w, c = 5, 120, 216, 256
    x = ops.to_act(torch.randn(n, h, w, c, device=DEV))
    L3 = ConvLayer.pack(torch.randn(c, c, 3, 3) * 0.02, torch.randn(c) * 0.1, None, 1, 1).to(DEV)
    y = ops.alloc_act(n, h, w, c, x.device)
    t_direct = timeit(lambda: ops.conv(x, L3, relu_out=True, out=y, out_act=True))
    # 16 transform-domain GEMMs: M = 16 * (h/2) * (w/2) * n = 518 400 rows of 256 channels
    tiles = n * (h // 2) * (w // 2)
    v = ops.to_act(torch.randn(16, tiles // 1080, 1080, c, device=DEV))            # [16, 30, 1080, 256]: M = 518 400
    L1 = ConvLayer.pack(torch.randn(c, c, 1, 1) * 0.05, None, None, 1, 0).to(DEV)
    prod = torch.empty((16, tiles // 1080, 1080, c), dtype=torch.float32, device=DEV)
    t_gemm = timeit(lambda: ops.conv(v, L1, out=prod))
    vf = torch.randn(16, tiles // 1080, 1080, c, device=DEV)
    va = ops.alloc_act(16, tiles // 1080, 1080, c, vf.device)
    t_vpass = timeit(lambda: ops.to_act(vf, out=va))
    t_unpack = timeit(lambda: ops.to_f32(va))
    small = torch.randn(n, h, w, c, device=DEV)
    sa = ops.alloc_act(n, h, w, c, small.device)
    t_small = timeit(lambda: ops.to_act(small, out=sa))
    bytes_v = 16 * tiles * c * 4
    rec = dict(layer="decoder up_8_4 out_conv 3x3 256->256, M = 129600", direct_us=round(t_direct, 1),
               direct_tflops=round(2 * n * h * w * c * c * 9 / t_direct / 1e6, 1),
               gemm16_us=round(t_gemm, 1), gemm16_tflops=round(2 * 16 * tiles * c * c / t_gemm / 1e6, 1),
               v_bytes=bytes_v, pack_pass_on_v_us=round(t_vpass, 1), pack_pass_gbs=round(2 * bytes_v / t_vpass / 1e3, 1),
               unpack_pass_on_v_us=round(t_unpack, 1), pack_pass_on_activation_us=round(t_small, 1),
               winograd_lower_bound_us=round(t_gemm + (t_vpass - t_small) / 2 + t_unpack / 2, 1),
               note="lower bound = gemm16 + half the extra pack-pass time of writing V instead of x (write side only) + half an unpack pass over the "
                    "products (read side only); transforms' arithmetic and the 4x4-patch gather not counted")
    rec["speedup_upper_bound"] = round(t_direct / rec["winograd_lower_bound_us"], 3)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()

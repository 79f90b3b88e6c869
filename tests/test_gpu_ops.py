"""Parity of every HIP kernel family with the CPU oracle / plain torch fp32 (needs an MI355X).
All calls go through the C ABI (mivos_amd.ops -> libmivos_hip.so)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mivos_amd import ops
from mivos_amd.ops import ConvLayer
from oracle import stm_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda:0"


def rel_err(got, ref):
    return float((got.double() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CONV_CASES = [
    # n, cin, cout, k, stride, pad, h, w, relu_in, relu_out, res, bn
    (1, 64, 64, 1, 1, 0, 30, 54, False, True, False, True),
    (2, 64, 256, 1, 1, 0, 17, 23, False, True, True, True),       # ragged M, residual
    (1, 256, 128, 1, 2, 0, 30, 54, False, False, False, True),    # strided 1x1 (downsample)
    (1, 128, 128, 3, 2, 1, 31, 37, False, True, False, True),     # 3x3 stride 2, odd size
    (3, 64, 64, 3, 1, 1, 20, 28, True, False, True, False),       # pre-activation, residual
    (1, 1024, 640, 3, 1, 1, 8, 10, False, False, False, False),   # KeyValue-like, K = 9216
    (1, 4, 64, 7, 2, 3, 64, 96, False, True, False, True),        # stem (3->4 padded)
    (2, 8, 64, 7, 2, 3, 48, 80, False, True, False, True),        # mask stem (5->8 padded)
    (1, 16, 32, 3, 1, 1, 48, 64, False, True, False, False),      # FusionNet conv1 (9->16 padded)
    (2, 32, 32, 3, 1, 1, 33, 47, False, True, True, False),       # FusionNet residual conv
    (2, 32, 1, 3, 1, 1, 33, 47, False, False, False, False),      # FusionNet head (Cout = 1)
    (2, 256, 1, 3, 1, 1, 30, 54, True, False, False, False),      # decoder.pred (Cout = 1, relu_in)
    (5, 256, 256, 3, 1, 1, 120, 216, True, False, True, False),   # largest decoder shape (128x128 tile path)
    (1, 512, 200, 3, 1, 1, 9, 11, False, False, False, False),    # Cout not a multiple of the tile
]


@pytest.fixture(params=["f32", "f16x3"])
def precision(request):
    """Both convolution back-ends: exact fp32 MFMA and error-compensated fp16 MFMA (3 products/term)."""
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, request.param
    yield request.param
    ops.CONV_PRECISION = old


@pytest.fixture(params=["f32", "f16x3", "f16x3-q128", "f16x3-q256", "f16x3-q256hf"])
def mem_precision(request):
    """The affinity kernels of the memory read: exact fp32 MFMA, error-compensated fp16 MFMA with 16 queries per wave,
    with 32 queries per wave (128 per workgroup, the long-memory kernel), with 8 waves of 32 (256 per workgroup, candidate
    regions in global scratch) and the latter's hi-first variant (hi x hi product first, lo products only where a bound of
    them cannot rule candidates out) - the long-memory kernels forced onto the small test shapes."""
    from mivos_amd import _lib
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, request.param.split("-")[0]
    old_min = _lib.load().mivos_memory_read_set_q128_min(0 if request.param.endswith("q128") else 1 << 40)
    old_256 = _lib.load().mivos_memory_read_set_q256_min(0 if "q256" in request.param else 1 << 60)
    old_hf = _lib.load().mivos_memory_read_set_hifirst(1 if request.param.endswith("hf") else 0)
    yield request.param
    ops.CONV_PRECISION = old
    _lib.load().mivos_memory_read_set_q128_min(old_min)
    _lib.load().mivos_memory_read_set_q256_min(old_256)
    _lib.load().mivos_memory_read_set_hifirst(old_hf)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c[:8])))
def test_conv2d_fused(case, precision):
    n, cin, cout, k, stride, pad, h, w, relu_in, relu_out, use_res, use_bn = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    bn = None
    if use_bn:
        bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
              torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    xin = F.relu(x) if relu_in else x
    ref = F.conv2d(xin.double(), wt.double(), b.double(), stride=stride, padding=pad)
    if bn is not None:
        ref = F.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0., 1e-5)
    res = None
    if use_res:
        res = torch.randn(n, cout, ref.shape[2], ref.shape[3], generator=g)
        ref = ref + res.double()
    if relu_out:
        ref = F.relu(ref)
    L = ConvLayer.pack(wt, b, bn, stride, pad).to(DEV)
    got = ops.conv(nhwc(x).to(DEV), L, relu_in=relu_in, relu_out=relu_out, res=None if res is None else nhwc(res).to(DEV))
    torch.cuda.synchronize()
    err = rel_err(got.cpu().permute(0, 3, 1, 2), ref)
    print(f"{precision}: rel err vs fp64 {err:.2e}")
    assert err < max(2e-6, 6e-8 * (cin * k * k) ** 0.5)   # fp32-class accuracy for both back-ends


def test_conv2d_strided_views_split_and_broadcast_residual(precision):
    g = torch.Generator().manual_seed(5)
    n, cin, h, w = 3, 64, 12, 14
    big = torch.randn(n, h, w, 160, generator=g).to(DEV)                  # input is a channel slice
    x = big[..., 32:96]
    wt = torch.randn(96, cin, 3, 3, generator=g) * 0.05
    b = torch.randn(96, generator=g) * 0.1
    res = torch.randn(1, 96, h, w, generator=g)                            # batch-1 residual, broadcast
    L = ConvLayer.pack(wt, b, None, 1, 1)
    L.split = 32
    L = L.to(DEV)
    bank_a = torch.zeros(n, 4, h, w, 32, device=DEV)                        # destination = slot 2 of a bank
    bank_b = torch.zeros(n, h, w, 100, device=DEV)                          # destination = channel slice
    ops.conv(x, L, res=nhwc(res).to(DEV), out=bank_a[:, 2], out2=bank_b[..., 10:74])
    ref = F.conv2d(x.cpu().permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=1) + res.double()
    got = torch.cat([bank_a[:, 2], bank_b[..., 10:74]], -1).cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 2e-6
    assert float(bank_a[:, [0, 1, 3]].abs().max()) == 0 and float(bank_b[..., :10].abs().max()) == 0 and float(bank_b[..., 74:].abs().max()) == 0


@pytest.mark.parametrize("shape", [(2, 64, 256, 21, 45), (1, 128, 640, 9, 33), (3, 32, 512, 8, 32)])
def test_conv3x3_direct_patch_kernel_ragged(shape, monkeypatch):
    """The LDS-patch-reuse 3x3 kernel (normally only picked for >= 200 tiles) on ragged images: partial 8x32
    tiles, several 32-channel slabs, Cout not a multiple of 256, pre-activation + residual + split."""
    monkeypatch.setenv("MIVOS_DIRECT3X3_MIN_TILES", "0")
    n, cin, cout, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g)
    ref = F.relu(F.conv2d(F.relu(x).double(), wt.double(), b.double(), padding=1) + res.double())
    L = ConvLayer.pack(wt, b, None, 1, 1)
    L.split = 128
    L = L.to(DEV)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f16x3"
    try:
        y1, y2 = ops.conv(nhwc(x).to(DEV), L, relu_in=True, relu_out=True, res=nhwc(res).to(DEV))
    finally:
        ops.CONV_PRECISION = old
    got = torch.cat([y1, y2], -1).cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 2e-6


DMA_CASES = [
    # n, cin, cout, k, stride, pad, h, w, relu_out, res ("", "f32", "act"), bn, out_act
    (1, 64, 64, 1, 1, 0, 30, 54, True, "", True, True),          # 128x64 tile
    (2, 64, 256, 1, 1, 0, 17, 23, True, "act", True, True),      # ragged M, SH32 residual, SH32 out (bottleneck conv3)
    (1, 256, 128, 1, 2, 0, 30, 54, False, "", True, True),       # strided 1x1 (downsample)
    (1, 128, 128, 3, 2, 1, 31, 37, True, "", True, True),        # 3x3 stride 2, odd size: border taps on every side
    (3, 64, 96, 3, 1, 1, 20, 28, False, "f32", False, False),    # Cout not a multiple of the tile, fp32 residual + output
    (1, 1024, 640, 3, 1, 1, 8, 10, False, "", False, False),     # KeyValue-like: K = 288 steps, split-K over the workspace
    (5, 256, 256, 3, 1, 1, 120, 216, False, "f32", False, False),  # largest decoder shape (128x256 tiles)
    (1, 512, 200, 3, 1, 1, 9, 11, True, "f32", False, True),     # Cout % 32 != 0: SH32 output refused, columns past Cout masked, split-K
    (1, 512, 192, 3, 1, 1, 9, 11, True, "act", False, True),     # split-K reduce with SH32 residual and SH32 output
    (2, 32, 160, 1, 1, 0, 5, 7, False, "", False, True),         # a single K step (prologue-only pipeline)
    (1, 64, 128, 3, 1, 1, 1, 1, False, "", False, True),         # 1x1 image: every tap but the centre reads the border
]


@pytest.mark.parametrize("case", DMA_CASES, ids=lambda c: "x".join(map(str, c[:8])))
def test_conv2d_lds_dma_path(case):
    """Precision 2 (conv_f16x3_dma.hip): SH32 activations in zero-bordered buffers in, SH32 or fp32 out, against fp64."""
    n, cin, cout, k, stride, pad, h, w, relu_out, res_kind, use_bn, out_act = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    bn = None
    if use_bn:
        bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
              torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad)
    if bn is not None:
        ref = F.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0., 1e-5)
    res = None
    if res_kind:
        res = torch.randn(n, cout, ref.shape[2], ref.shape[3], generator=g)
        ref = ref + res.double()
    if relu_out:
        ref = F.relu(ref)
    L = ConvLayer.pack(wt, b, bn, stride, pad).to(DEV)
    xa = ops.to_act(nhwc(x).to(DEV))
    assert float((ops.to_f32(xa).cpu() - nhwc(x)).abs().max()) <= 2.0 ** -21 * float(x.abs().max())      # hi + lo keeps 22 bits
    r = None
    if res_kind == "f32":
        r = nhwc(res).to(DEV)
    elif res_kind == "act":
        r = ops.to_act(nhwc(res).to(DEV))
    if out_act and cout % 32:
        with pytest.raises(ops.MivosHipError):
            ops.conv(xa, L, relu_out=relu_out, res=r, out_act=True)
        out_act = False
    got = ops.conv(xa, L, relu_out=relu_out, res=r, out_act=out_act)
    if out_act:
        border = got.buf.clone()
        border[:, 1:-1, 1:-1] = 0
        assert float(border.abs().max()) == 0                      # the zero border survives the epilogue
        got = ops.to_f32(got)
    torch.cuda.synchronize()
    err = rel_err(got.cpu().permute(0, 3, 1, 2), ref)
    print(f"lds-dma: rel err vs fp64 {err:.2e}")
    assert err < max(2e-6, 6e-8 * (cin * k * k) ** 0.5)


def test_sh32_pack_kernels_match_the_cpu_restatement_bitwise():
    """mivos_pack_activation_sh32 / mivos_unpack_activation_sh32 / mivos_pack_weights_f16x3_dma vs oracle/sh32.py."""
    from oracle import sh32
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 9, 13, 96, generator=g) * 3.0
    for relu in (False, True):
        a = ops.to_act(x.to(DEV), relu=relu)
        assert torch.equal(a.buf.cpu().view(torch.int32), sh32.pack_activation(x, relu=relu).view(torch.int32))     # incl. zero border
        assert torch.equal(ops.to_f32(a).cpu(), sh32.unpack_activation(a.buf.cpu()))
    for cout, k, cin in [(40, 3, 64), (130, 1, 96)]:
        w = torch.randn(cout, cin, k, k, generator=g) * 0.05
        L = ConvLayer.pack(w, None, None, 1, k // 2).to(DEV)
        wd, _ = L.dma()
        mult, _ = L._f16x3_scale()
        ref = sh32.pack_weights_dma(L.w.cpu(), mult)
        assert wd.numel() == ref.numel() and torch.equal(wd.cpu(), ref)


def test_conv2d_lds_dma_matches_register_staged_kernel_bitwise():
    """Same products, same accumulation order: the two f16x3 back-ends agree bit for bit (no split-K on either side)."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(8, 60, 108, 128, generator=g).to(DEV)         # 405 tiles of 128x128: neither back-end splits K
    L = ConvLayer.pack(torch.randn(128, 128, 3, 3, generator=g) * 0.03, torch.randn(128, generator=g) * 0.1, None, 1, 1).to(DEV)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f16x3"
    try:
        a = ops.conv(x, L, relu_out=True)
        b = ops.conv(ops.to_act(x), L, relu_out=True)
    finally:
        ops.CONV_PRECISION = old
    # x itself is rounded to hi + lo by to_act exactly as the register-staged kernel does while staging
    assert torch.equal(a, b)


FOLD_CASES = [
    # n, cin, cout, k, stride, h, w, relu_out, res ("", "f32", "act"), bn, out_act     -- shapes whose grids split K (<= 1/4 of the workgroup slots)
    (5, 256, 256, 3, 1, 30, 54, True, "", True, True),          # the M = 8100 bottleneck 3x3 of the memory encoder: 128 workgroups, 72 steps -> 4 slices
    (5, 256, 256, 3, 2, 60, 108, True, "", True, True),         # its stride-2 sibling
    (1, 1024, 256, 1, 1, 30, 54, True, "", True, True),         # one frame: 26 workgroups, 32 steps -> 4 slices
    (1, 512, 256, 3, 1, 30, 54, True, "act", True, True),       # SH32 residual + SH32 output through the folded epilogue (26 workgroups, 144 steps -> 8 slices)
    (1, 512, 192, 3, 1, 9, 11, True, "act", False, True),       # ragged everything, last slice shorter (144 steps / 8 slices of 18)
    (1, 512, 200, 3, 1, 9, 11, False, "f32", False, False),     # fp32 output and residual, Cout % 32 != 0
    (2, 1024, 640, 3, 1, 8, 10, False, "", False, False),       # two destinations would need split < Cout: single fp32 destination here, 288 steps
]


@pytest.mark.parametrize("case", FOLD_CASES, ids=lambda c: "x".join(map(str, c[:7])))
def test_conv2d_split_k_is_independent_of_chip_share_bitwise(case):
    """Round 6: a layer that splits K gives the SAME BITS whether its slices run as separate workgroups + splitk_reduce_kernel (one launch stream:
    mivos_conv_desc.chip_share <= 1) or one after the other inside one workgroup (conv_f16x3_pp_kernel<..., FOLD>: chip_share > 1, what the lanes of a
    suite / the two passes of an interaction say).  Round 5's share-aware slice COUNT changed the fp32 summation order, and with it a clip's masks."""
    n, cin, cout, k, stride, h, w, relu_out, res_kind, use_bn, out_act = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5) if use_bn else None
    L = ConvLayer.pack(wt, b, bn, stride, k // 2).to(DEV)
    xa = ops.to_act(nhwc(x).to(DEV))
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    r = None
    if res_kind:
        res = torch.randn(n, ho, wo, cout, generator=g).to(DEV)
        r = res if res_kind == "f32" else ops.to_act(res)
    from mivos_amd import _lib
    lib = _lib.load()
    outs = []
    old_mode = lib.mivos_conv2d_set_fold_mode(-1)
    try:
        # (share, fold mode): the default rule with one / two / three streams, then every split layer forced through the split + reduce route and through
        # the folded route (small grids fold only when forced: mivos_conv2d_set_fold_mode)
        for share, mode in ((1, 1), (2, 1), (3, 1), (2, 0), (1, 2)):
            lib.mivos_conv2d_set_fold_mode(mode)
            with ops.chip_share(share):
                y = ops.conv(xa, L, relu_out=relu_out, res=r, out_act=out_act)
            outs.append((y.buf if out_act else y).clone())
    finally:
        lib.mivos_conv2d_set_fold_mode(old_mode)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0].view(torch.int32), o.view(torch.int32)) for o in outs[1:])
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=k // 2)
    if bn is not None:
        ref = F.batch_norm(ref, bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0., 1e-5)
    if res_kind:
        ref = ref + res.cpu().permute(0, 3, 1, 2).double()
    if relu_out:
        ref = F.relu(ref)
    got = ops.to_f32(ops.Act(outs[1], n, ho, wo, cout)) if out_act else outs[1]
    assert rel_err(got.cpu().permute(0, 3, 1, 2), ref) < max(2e-6, 6e-8 * (cin * k * k) ** 0.5)


def test_conv2d_split_k_dual_destination_is_independent_of_chip_share_bitwise():
    """KeyValue of a one-object clip (1024 -> 128 + 512 at 30 x 54: 65 tiles, 288 K steps -> 7 slices) writes TWO fp32 destinations from one GEMM; with other streams
    on the chip its slices run folded (>= 64 tiles).  Same bits on both destinations for every (share, fold mode)."""
    from mivos_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 30, 54, 1024, generator=g).to(DEV)
    wt, b = torch.randn(640, 1024, 3, 3, generator=g) * 0.01, torch.randn(640, generator=g) * 0.1
    L = ConvLayer.pack(wt, b, None, 1, 1)
    L.split = 128
    L = L.to(DEV)
    xa = ops.to_act(x)
    outs = []
    old_mode = lib.mivos_conv2d_set_fold_mode(-1)
    try:
        for share, mode in ((1, 1), (2, 1), (2, 0), (1, 2)):
            lib.mivos_conv2d_set_fold_mode(mode)
            with ops.chip_share(share):
                y1, y2 = ops.conv(xa, L)
            outs.append((y1.clone(), y2.clone()))
    finally:
        lib.mivos_conv2d_set_fold_mode(old_mode)
    torch.cuda.synchronize()
    for y1, y2 in outs[1:]:
        assert torch.equal(outs[0][0].view(torch.int32), y1.view(torch.int32)) and torch.equal(outs[0][1].view(torch.int32), y2.view(torch.int32))
    ref = F.conv2d(x.cpu().permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=1)
    assert rel_err(torch.cat(outs[1], -1).cpu().permute(0, 3, 1, 2), ref) < 6e-8 * (1024 * 9) ** 0.5


def test_conv2d_lds_dma_dual_destination_and_batch_slices():
    """KeyValue-style split into two fp32 destinations from an Act input; Act batch slices keep their borders."""
    g = torch.Generator().manual_seed(3)
    n, h, w = 3, 9, 12
    x = torch.randn(n, h, w, 64, generator=g).to(DEV)
    wt, b = torch.randn(96, 64, 3, 3, generator=g) * 0.05, torch.randn(96, generator=g) * 0.1
    L = ConvLayer.pack(wt, b, None, 1, 1)
    L.split = 32
    L = L.to(DEV)
    xa = ops.to_act(x)
    y1, y2 = ops.conv(xa, L)
    ref = F.conv2d(x.cpu().permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=1)
    assert rel_err(torch.cat([y1, y2], -1).cpu().permute(0, 3, 1, 2), ref) < 2e-6
    z1, z2 = ops.conv(xa[1:2], L)
    assert torch.equal(z1, y1[1:2]) and torch.equal(z2, y2[1:2])


def test_conv2d_lds_dma_rejects_what_it_cannot_do():
    L = ConvLayer.pack(torch.randn(64, 64, 3, 3), None, None, 1, 1).to(DEV)
    xa = ops.to_act(torch.randn(1, 8, 8, 64, device=DEV))
    with pytest.raises(ops.MivosHipError):
        ops.conv(xa, L, relu_in=True)                                       # a DMA-staged operand cannot be modified on load
    with pytest.raises(ops.MivosHipError):
        ops.conv(torch.randn(1, 8, 8, 64, device=DEV), L, out_act=True)     # SH32 outputs come from the LDS-DMA kernels only
    with pytest.raises(ops.MivosHipError):
        ops.alloc_act(1, 8, 8, 48, torch.device(DEV))                       # channels % 32


def test_conv_rejects_bad_arguments():
    L = ConvLayer.pack(torch.randn(8, 12, 3, 3), None, None, 1, 1).to(DEV)
    with pytest.raises(ops.MivosHipError):
        ops.conv(torch.zeros(1, 8, 8, 12, device=DEV), L)                   # Cin not a power of two
    with pytest.raises(ops.MivosHipError):
        ops.conv(torch.zeros(1, 8, 8, 16), ConvLayer.pack(torch.randn(8, 16, 3, 3), None, None, 1, 1))  # CPU tensor


@pytest.mark.parametrize("B,H,W", [(1, 8, 30), (1, 9, 31), (2, 37, 61), (1, 48, 80), (5, 480, 864), (1, 16, 1000)])
def test_fusion_resblock_one_launch_vs_two_convolutions(B, H, W):
    """mivos_fusion_resblock: relu(x + conv_b(relu(conv_a(x)))) with the intermediate in LDS (csrc/fusion_net.hip), against the
    two fused-epilogue convolutions of the layer-by-layer path (same f16x3 products in the same order: agreement at rounding
    level, any tile / halo / zero-padding indexing error is O(1)) and against fp64 torch.  Sizes: one tile exactly, one pixel
    more than a tile each way, ragged, the benchmark's batch, a wide strip (34 tiles across)."""
    g = torch.Generator().manual_seed(B * 7919 + H * 31 + W)
    x = torch.randn(B, 32, H, W, generator=g)
    wa, wb = (torch.randn(32, 32, 3, 3, generator=g) * (2.0 / 288) ** 0.5 for _ in range(2))
    ba, bb = (torch.randn(32, generator=g) * 0.1 for _ in range(2))
    La, Lb = ConvLayer.pack(wa, ba, None, 1, 1).to(DEV), ConvLayer.pack(wb, bb, None, 1, 1).to(DEV)
    xd = nhwc(x).to(DEV)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f16x3"
    try:
        got = ops.fusion_resblock(xd, La, Lb)
        two = ops.conv(ops.conv(xd, La, relu_out=True), Lb, res=xd, relu_out=True)
        nobias = ops.fusion_resblock(xd, ConvLayer.pack(wa, None, None, 1, 1).to(DEV), ConvLayer.pack(wb, None, None, 1, 1).to(DEV))
    finally:
        ops.CONV_PRECISION = old
    ref = F.relu(x.double() + F.conv2d(F.relu(F.conv2d(x.double(), wa.double(), ba.double(), padding=1)), wb.double(), bb.double(), padding=1))
    ref0 = F.relu(x.double() + F.conv2d(F.relu(F.conv2d(x.double(), wa.double(), None, padding=1)), wb.double(), None, padding=1))
    d2 = float((got - two).abs().max())
    print(f"fusion_resblock [{B}x{H}x{W}]: vs two launches {d2:.2e}, vs fp64 {rel_err(got.cpu().permute(0, 3, 1, 2), ref):.2e} (two launches: {rel_err(two.cpu().permute(0, 3, 1, 2), ref):.2e})")
    assert got.shape == (B, H, W, 32) and d2 < 1e-5
    assert rel_err(got.cpu().permute(0, 3, 1, 2), ref) < 3e-6 and rel_err(nobias.cpu().permute(0, 3, 1, 2), ref0) < 3e-6
    with pytest.raises(AssertionError):
        ops.fusion_resblock(xd, La, Lb, out=xd)                                # in place is refused


@pytest.mark.parametrize("B,H,W", [(1, 8, 32), (3, 37, 61), (5, 480, 864)])
def test_fusion_conv1_from_planes(B, H, W):
    """mivos_fusion_conv1_planes (conv1 of FusionNet gathering its nine planar inputs itself, fusion_net.py:38-40) against the
    interleave + direct-convolution path: shared image planes (batch stride 0), per-object planes, a strided two-plane
    tensor and two constant planes - the operands InferenceCore.fuse_one_frame passes."""
    g = torch.Generator().manual_seed(B * 13 + W)
    P = H * W
    im = torch.randn(3, H, W, generator=g).to(DEV)
    s1, s2 = torch.rand(B, 1, H, W, generator=g).to(DEV), torch.rand(B, 1, H, W, generator=g).to(DEV)
    attn = torch.rand(2 * B, H, W, generator=g).to(DEV)
    w, b = torch.randn(32, 9, 3, 3, generator=g) * (2.0 / 81) ** 0.5, torch.randn(32, generator=g) * 0.1
    L = ConvLayer.pack(w, b, None, 1, 1, cin_pad=16).to(DEV)
    imf, atf = im.reshape(-1), attn.reshape(-1)
    planes = [(imf[c * P:], 0) for c in range(3)] + [(s1, P), (s2, P)] + [(atf[c * P:], 2 * P) for c in range(2)] + [(0.25, 0), (0.75, 0)]
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f16x3"
    try:
        got = ops.fusion_conv1_planes(planes, (B, H, W), L)
        ref = ops.conv(ops.interleave(planes, B, P, 16, im.device).view(B, H, W, 16), L, relu_out=True)
    finally:
        ops.CONV_PRECISION = old
    x = torch.cat([im.cpu().expand(B, -1, -1, -1), s1.cpu(), s2.cpu(), attn.cpu().view(B, 2, H, W), torch.full((B, 1, H, W), 0.25), torch.full((B, 1, H, W), 0.75)], 1)
    ref64 = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    assert got.shape == (B, H, W, 32) and float((got - ref).abs().max()) < 1e-5
    assert rel_err(got.cpu().permute(0, 3, 1, 2), ref64) < 3e-6


@pytest.mark.parametrize("B,H,W", [(1, 8, 32), (2, 37, 61), (5, 480, 864)])
def test_fusion_head_exact_fp32(B, H, W):
    """mivos_fusion_head (final_conv 32 -> 1, fusion_net.py:49) in exact fp32 FMA vs fp64 torch and vs the f16x3 projection +
    tap-sum path it replaces."""
    g = torch.Generator().manual_seed(B * 31 + W)
    x = torch.randn(B, 32, H, W, generator=g)
    w, b = torch.randn(1, 32, 3, 3, generator=g) * (2.0 / 288) ** 0.5, torch.randn(1, generator=g)
    L = ConvLayer.pack(w, b, None, 1, 1).to(DEV)
    xd = nhwc(x).to(DEV)
    got = ops.fusion_head(xd, L)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f16x3"
    try:
        proj = ops.conv(xd, L)
    finally:
        ops.CONV_PRECISION = old
    assert got.shape == (B, H, W, 1) and rel_err(got.cpu().permute(0, 3, 1, 2), ref) < 1e-6
    assert float((got - proj).abs().max()) < 1e-5 * float(ref.abs().max())
    nb = ops.fusion_head(xd, ConvLayer.pack(w, None, None, 1, 1).to(DEV))
    assert rel_err(nb.cpu().permute(0, 3, 1, 2), ref - b.double()) < 1e-6


@pytest.mark.parametrize("N,n_planes,cin,H,W", [(1, 3, 4, 64, 96), (2, 5, 8, 50, 70), (5, 5, 8, 480, 864), (8, 3, 4, 128, 160), (1, 5, 8, 17, 33),
                                                 (1, 5, 8, 480, 864), (3, 3, 4, 480, 864)])
def test_stem_from_planes(N, n_planes, cin, H, W):
    """mivos_stem7x7s2_planes (7x7 / 2 / pad 3 conv + BN + ReLU straight from planar inputs: the stems of the query encoder - 3
    planes, Cin padded to 4 - and of the mask encoder - frame planes shared by all objects + per-object mask / others planes) vs
    the interleave + implicit-GEMM path it replaces and vs fp64 torch; even, odd and tiny sizes (tiles cut by the image edge)."""
    g = torch.Generator().manual_seed(N * 100 + H)
    P = H * W
    frame = torch.randn(3, H, W, generator=g).to(DEV)
    extra = torch.rand(N, max(n_planes - 3, 1), H, W, generator=g).to(DEV)
    w = torch.randn(64, n_planes, 7, 7, generator=g) * (2.0 / (49 * n_planes)) ** 0.5
    bn = (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5)
    L = ConvLayer.pack(w, None, bn, 2, 3, cin_pad=cin).to(DEV)
    if n_planes == 3:
        batch = torch.randn(N, 3, H, W, generator=g).to(DEV)                      # N different frames (query batches)
        flat = batch.reshape(-1)
        planes = [(flat[c * P:], 3 * P) for c in range(3)]
        x = batch.cpu()
    else:
        ef = extra.reshape(-1)
        planes = [(frame[c], 0) for c in range(3)] + [(ef[j * P:], (n_planes - 3) * P) for j in range(n_planes - 3)]
        x = torch.cat([frame.cpu().expand(N, -1, -1, -1), extra.cpu()], 1)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f16x3"
    try:
        got = ops.stem_planes(planes, N, H, W, L)
        ref = ops.conv(ops.interleave(planes, N, P, cin, frame.device).view(N, H, W, cin), L, relu_out=True)
    finally:
        ops.CONV_PRECISION = old
    s = (bn[0] / torch.sqrt(bn[3] + 1e-5)).double()
    ref64 = F.relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=3) * s[None, :, None, None] + (bn[1].double() - bn[2].double() * s)[None, :, None, None])
    assert got.shape == ref.shape == (N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64)
    e, r = rel_err(got.cpu().permute(0, 3, 1, 2), ref64), rel_err(ref.cpu().permute(0, 3, 1, 2), ref64)
    print(f"stem [{N}x{n_planes}x{H}x{W}]: vs fp64 {e:.2e} (implicit-GEMM path: {r:.2e}), vs that path {float((got - ref).abs().max()):.2e}")
    assert e < 3e-6 and float((got - ref).abs().max()) < 2e-5 * float(ref64.abs().max())


def test_maxpool_and_upsample_add():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 37, 45, generator=g)
    got = ops.maxpool3x3s2(nhwc(x).to(DEV)).cpu().permute(0, 3, 1, 2)
    assert torch.equal(got, F.max_pool2d(x, 3, 2, 1))
    up, skip = torch.randn(3, 32, 15, 27, generator=g), torch.randn(1, 32, 30, 54, generator=g)
    ref = skip + F.interpolate(up, scale_factor=2, mode="bilinear", align_corners=False)
    got = ops.upsample2x_add(nhwc(skip).to(DEV), nhwc(up).to(DEV)).cpu().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("N,h,w,C,skip_n", [(5, 60, 108, 256, 1), (3, 15, 27, 32, 3), (1, 7, 5, 64, 1), (2, 30, 54, 512, 1)])
def test_upsample_add_acts_and_maxpool_act_vs_the_fp32_kernels(N, h, w, C, skip_n):
    """The SH32-writing forms the decoder / encoder trunks use (mivos_upsample2x_add_multi: raw + relu Acts; mivos_maxpool3x3s2_sh32), whose
    workgroups walk XCD-contiguous runs of the output, against the dense fp32 kernels (checked against torch above): the benchmark's decoder
    shape, ragged sizes with fewer elements than the grid, per-sample and broadcast skip tensors.  x = hi + lo carries 22 bits."""
    g = torch.Generator().manual_seed(N * 100 + h)
    up = torch.randn(N, h, w, C, generator=g).to(DEV)
    skip = torch.randn(skip_n, 2 * h, 2 * w, C, generator=g).to(DEV)
    ref = ops.upsample2x_add(skip, up)
    raw, rel = ops.upsample2x_add_acts(skip, up, "test.up")
    tol = 2.0 ** -21 * float(ref.abs().max())
    assert float((ops.to_f32(raw) - ref).abs().max()) <= tol and float((ops.to_f32(rel) - ref.clamp(min=0)).abs().max()) <= tol
    assert float(raw.buf[:, 0].abs().max()) == 0 and float(raw.buf[:, :, 0].abs().max()) == 0 and float(raw.buf[:, -1].abs().max()) == 0     # border untouched
    x = torch.randn(N, 2 * h, 2 * w, C, generator=g).to(DEV)
    a = ops.maxpool3x3s2(x, act_tag="test.pool", as_act=True)
    want = ops.maxpool3x3s2(x)
    assert a.shape == tuple(want.shape) and float((ops.to_f32(a) - want).abs().max()) <= 2.0 ** -21 * float(want.abs().max())


def test_resize_area_sigmoid():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 30, 54, generator=g)
    for H, W in ((120, 216), (480, 864)):
        ref = F.interpolate(x[None], size=(H, W), mode="bilinear", align_corners=False)[0]
        assert float((ops.resize_bilinear(x.to(DEV), H, W).cpu() - ref).abs().max()) < 1e-5
    ref = torch.sigmoid(F.interpolate(x[None], scale_factor=4, mode="bilinear", align_corners=False)[0])
    assert float((ops.resize_bilinear(x.to(DEV), 120, 216, act=1).cpu() - ref).abs().max()) < 1e-6
    m = torch.rand(4, 64, 96, generator=g)
    ref = F.interpolate(m[None], size=(4, 6), mode="area")[0]
    assert float((ops.area_pool16(m.to(DEV)).cpu() - ref).abs().max()) < 1e-6
    assert float((ops.sigmoid(x.to(DEV)).cpu() - torch.sigmoid(x)).abs().max()) < 1e-6


def test_aggregate_argmax_diff_others(golden_dir):
    with np.load(os.path.join(golden_dir, "ops_small.npz")) as z:
        p, soft, hard, sbg = (torch.from_numpy(z[k]) for k in ("ag_in", "ag_soft", "ag_hard", "ag_sbg"))
    pd = p.to(DEV)
    assert float((ops.aggregate(pd, keep_bg=True).cpu() - soft).abs().max()) < 2e-6
    assert float((ops.aggregate(pd, keep_bg=True, hard=True).cpu() - hard).abs().max()) < 2e-6
    assert float((ops.aggregate(pd, keep_bg=True, soft_bg=False).cpu() - sbg).abs().max()) < 2e-6
    assert float((ops.aggregate(pd, keep_bg=False).cpu() - soft[1:]).abs().max()) < 2e-6
    g = torch.Generator().manual_seed(3)
    many = torch.rand(40, 1, 16, 24, generator=g) * 0.2       # more objects than the register-cached variant holds (32)
    for hard in (False, True):
        got = ops.aggregate(many.to(DEV), keep_bg=True, hard=hard).cpu()
        ref = O.aggregate_wbg(many, True, hard=hard)
        # hard: logits x 1000, so a 1-ulp difference of logf moves exp(l - max) by 1e-4 relative; the winner must be the same
        assert got.shape == (41, 1, 16, 24) and float((got - ref).abs().max()) < (5e-4 if hard else 2e-6)
        assert torch.equal(got.argmax(0), ref.argmax(0))
    prob = torch.rand(4, 7, 1, 16, 24, generator=g)
    prob[2, 3] = prob[1, 3]                                   # ties: first index must win
    got = ops.argmax_u8(prob.to(DEV).view(4, -1)).cpu().view(7, 1, 16, 24)
    assert torch.equal(got.long(), torch.argmax(prob, dim=0))
    mask, old = (torch.rand(3, 1, 16, 24, generator=g) > 0.5).float(), torch.rand(3, 1, 16, 24, generator=g)
    pos, neg = ops.mask_diff(mask.to(DEV), old.to(DEV))
    assert torch.equal(pos.cpu(), (mask - old).clamp(0, 1)) and torch.equal(neg.cpu(), (old - mask).clamp(0, 1))
    masks = torch.rand(4, 1, 8, 12, generator=g)
    ref = torch.cat([masks[[j for j in range(4) if j != i]].sum(0, keepdim=True) for i in range(4)], 0)
    assert float((ops.mask_others(masks.to(DEV)).cpu() - ref).abs().max()) < 1e-6


def test_interleave():
    g = torch.Generator().manual_seed(4)
    frame, masks = torch.randn(1, 3, 8, 12, generator=g).to(DEV), torch.rand(2, 1, 8, 12, generator=g).to(DEV)
    P = 96
    out = ops.interleave([(frame[0, c], 0) for c in range(3)] + [(masks, P), (0.25, 0)], 2, P, 8, frame.device)
    ref = torch.cat([frame.expand(2, -1, -1, -1), masks, torch.full_like(masks, 0.25), torch.zeros(2, 3, 8, 12, device=DEV)], 1)
    assert torch.equal(out.view(2, 8, 12, 8).permute(0, 3, 1, 2), ref)


# ------------------------------------------------------------------ memory read

def _mem_case(T, h, w, K, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    mk = torch.randn(K, 128, T, h, w, generator=g) * scale
    mv = torch.randn(K, 512, T, h, w, generator=g)
    qk = torch.randn(1, 128, h, w, generator=g) * scale
    return mk, mv, qk


def _run_mem(mk, mv, qk, top_k):
    K, _, T, h, w = mk.shape
    keys = mk.permute(0, 2, 3, 4, 1).reshape(K, T * h * w, 128).contiguous().to(DEV)
    vals = mv.permute(0, 2, 3, 4, 1).reshape(K, T * h * w, 512).contiguous().to(DEV)
    q = qk.permute(0, 2, 3, 1).reshape(h * w, 128).contiguous().to(DEV)
    out = ops.memory_read(keys, vals, q, top_k)
    idx, wgt = ops.memory_read_indices(keys, q, top_k)
    return out.cpu().view(K, h, w, 512).permute(0, 3, 1, 2), idx.cpu(), wgt.cpu()


def test_memory_read_golden(golden_dir, mem_precision):
    with np.load(os.path.join(golden_dir, "ops_small.npz")) as z:
        mk, mv, qk, ref = (torch.from_numpy(z[k]) for k in ("mr_mk", "mr_mv", "mr_qk", "mr_out"))
    got, _, _ = _run_mem(mk, mv, qk, 20)
    assert float((got - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("T,h,w,K,top_k", [(1, 8, 10, 1, 50), (3, 9, 13, 2, 50), (5, 30, 54, 1, 20), (5, 30, 54, 3, 50), (23, 30, 54, 1, 50),
                                           (40, 8, 10, 1, 50),      # one 64-query stream cut into many segments (10 lists to merge)
                                           (7, 30, 54, 5, 50),      # the benchmark's shape: runs that cross stream boundaries
                                           (2, 68, 120, 1, 64)])    # 1080p grid, largest supported k
def test_memory_read_vs_oracle(T, h, w, K, top_k, mem_precision):
    """Readout and exact top-k membership, for the exact fp32 MFMA affinity and for the error-compensated fp16 one (whose
    scores are as close to the exact product as torch's own fp32 ones: tests/test_host_logic.py).  A query whose k-th and (k+1)-th scores tie within fp32 rounding (the case
    T=23 holds one with a margin of exactly 0 in torch's own fp32 affinity) may legitimately resolve either way -
    torch.topk leaves ties unspecified and the summation order of the 128-term dot product is implementation defined -
    so values and index sets are compared on the queries with a clear margin (all but <= 1 % / two of them)."""
    mk, mv, qk = _mem_case(T, h, w, K, seed=T * 100 + K)
    got, idx, wgt = _run_mem(mk, mv, qk, top_k)
    for o in range(K):
        ref = O.memory_read(mk[o:o + 1], mv[o:o + 1], qk, top_k)
        a = O.affinity(mk[o:o + 1], qk)[0]                      # [THW, HW]
        vals, ridx = torch.topk(a, min(top_k + 1, a.shape[0]), dim=0)
        clear = (vals[top_k - 1] - vals[top_k]) > 1e-5 if a.shape[0] > top_k else torch.ones(a.shape[1], dtype=torch.bool)
        d = (got[o] - ref[0]).abs().amax(0).reshape(-1)         # per query, max over the 512 channels
        assert float(d[clear].max()) < 2e-4 and int((~clear).sum()) <= max(2, clear.numel() // 100)
        assert float(d.max()) < 0.5                             # a flipped tie swaps ONE neighbour of weight ~1/k
        got_sets = torch.sort(idx[o].long(), dim=1)[0]          # [HW, k]
        ref_sets = torch.sort(ridx[:top_k].t(), dim=1)[0]
        same = (got_sets == ref_sets).all(dim=1)
        assert bool(same[clear].all())
        clear1 = clear & ((vals[0] - vals[1]) > 1e-5)             # best first, where ranks 1 and 2 do not tie within fp32 rounding
        assert torch.equal(idx[o][:, 0].long()[clear1], ridx[0][clear1]) and int(clear1.sum()) >= int(clear.sum()) - max(2, clear.numel() // 100)
        assert float((wgt[o].sum(1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("frames,scale", [(50, 1.0), (150, 1.7)])
def test_memory_read_deep_bank_1080p_vs_chunked_oracle(frames, scale):
    """BASELINE config 5's regime (1080x1920 -> 68x120 = 8160 positions per frame, K = 3, top-50, banks of 8160 x {50, 150}
    = 408 k / 1.22 M positions per object) with the DEFAULT kernel selection (memread_select256_kernel from 200 k positions)
    and InferenceCore's bank geometry: `bank[:, :n]` views of pre-allocated [K, slots, h, w, C] banks whose object strides
    exceed 2^31 BYTES for keys, split keys and values.  Checker: oracle/chunked_read.py (the reference's affinity -> topk ->
    softmax -> readout, 256 queries at a time, in fp64 and in fp32, as plain torch on this GPU - the materialised affinity
    would be 40 - 120 GB).  Exact index sets on every query whose rank-50/51 margin (fp64) is clear, readout < 2e-4."""
    from mivos_amd import _lib
    from oracle import chunked_read as CR
    K, h, w, top_k = 3, 68, 120, 50
    hw = h * w
    g = torch.Generator(device=DEV).manual_seed(7000 + frames)
    kbank = torch.empty((K, 530, h, w, 128), dtype=torch.float32, device=DEV)     # object stride 2.21 GB
    vbank = torch.empty((K, 264, h, w, 512), dtype=torch.float32, device=DEV)     # object stride 4.41 GB
    sbank = torch.empty_like(kbank)
    assert kbank.stride(0) * 4 > 2 ** 31 and vbank.stride(0) * 4 > 2 ** 32
    for o in range(K):                                                             # (per object: bounded temporaries)
        kbank[o, :frames] = torch.randn((frames, h, w, 128), generator=g, device=DEV) * scale
        vbank[o, :frames] = torch.randn((frames, h, w, 512), generator=g, device=DEV)
    q = torch.randn((hw, 128), generator=g, device=DEV) * scale
    keys, vals = kbank[:, :frames].reshape(K, frames * hw, 128), vbank[:, :frames].reshape(K, frames * hw, 512)
    assert keys.data_ptr() == kbank.data_ptr() and vals.data_ptr() == vbank.data_ptr()          # views, not copies
    plan = (C.c_int32 * 8)()
    assert _lib.load().mivos_memory_read_plan(K, frames * hw, hw, top_k, 1, plan) == 0
    assert plan[6] in (128, 256)                                                  # a long-memory kernel is what runs by default
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f16x3"
    try:
        for t in range(0, frames, 25):                                            # the bank is split slot by slot, like do_pass
            ops.split_keys(kbank[:, t:t + 25], sbank[:, t:t + 25])
        ks = sbank[:, :frames].reshape(K, frames * hw, 128)
        got = ops.memory_read(keys, vals, q, top_k, keys_split=ks)
        idx, wgt = ops.memory_read_indices(keys, q, top_k, keys_split=ks)
        old256 = _lib.load().mivos_memory_read_set_q256_min(0)                      # the 256-query kernel (global candidate regions)
        try:
            assert _lib.load().mivos_memory_read_plan(K, frames * hw, hw, top_k, 1, plan) == 0 and plan[6] == 256
            got256 = ops.memory_read(keys, vals, q, top_k, keys_split=ks)
            idx256, _ = ops.memory_read_indices(keys, q, top_k, keys_split=ks)
            oldhf = _lib.load().mivos_memory_read_set_hifirst(1)                    # ... and its hi-first variant
            try:
                got_hf = ops.memory_read(keys, vals, q, top_k, keys_split=ks)
                idx_hf, _ = ops.memory_read_indices(keys, q, top_k, keys_split=ks)
            finally:
                _lib.load().mivos_memory_read_set_hifirst(oldhf)
        finally:
            _lib.load().mivos_memory_read_set_q256_min(old256)
        ops.CONV_PRECISION = "f32"
        got32 = ops.memory_read(keys, vals, q, top_k)                              # exact fp32 MFMA kernel, same geometry
        idx32, _ = ops.memory_read_indices(keys, q, top_k)
    finally:
        ops.CONV_PRECISION = old
    r64 = CR.memory_read_rows(keys, vals, q, top_k, dtype=torch.float64)
    r32 = CR.memory_read_rows(keys, vals, q, top_k, dtype=torch.float32)
    clear = r64["margin"] > 1e-5                                                  # [K, n_q]
    ref_sets = torch.sort(r64["idx"], dim=2)[0]
    for name, o_got, o_idx in (("f16x3/default", got, idx), ("f16x3/select256", got256, idx256), ("f16x3/select256 hi-first", got_hf, idx_hf), ("f32", got32, idx32)):
        d64 = (o_got.double() - r64["readout"]).abs().amax(2)
        d32 = (o_got - r32["readout"]).abs().amax(2)
        same = (torch.sort(o_idx.long(), dim=2)[0] == ref_sets).all(dim=2)
        print(f"deep bank T={frames} [{name}]: n_mem {frames * hw}, unclear queries {int((~clear).sum())} of {clear.numel()}, "
              f"readout max|d| vs fp64 {float(d64[clear].max()):.2e} / vs fp32 oracle {float(d32[clear].max()):.2e}, "
              f"oracle fp32 vs fp64 {float((r32['readout'].double() - r64['readout']).abs().amax(2)[clear].max()):.2e}, "
              f"index sets equal on {int(same[clear].sum())} of {int(clear.sum())} clear queries")
        assert int((~clear).sum()) <= clear.numel() // 100
        assert bool(same[clear].all())
        # best first - where the fp64 scores of rank 1 and rank 2 are further apart than fp32 rounding of a 128-term sum
        # (the order INSIDE the selected set moves neither the set nor, beyond 1e-6, the weights)
        top_clear = clear & ((r64["weights"][..., 0] / r64["weights"][..., 1]).log() > 1e-5)
        assert int(top_clear.sum()) > 0.99 * clear.numel()
        assert torch.equal(o_idx[..., 0].long()[top_clear], r64["idx"][..., 0][top_clear])
        assert float(d64[clear].max()) < 2e-4 and float(d32[clear].max()) < 2e-4
        assert float(d64.max()) < 0.5                                             # an unclear query swaps ONE neighbour of weight ~1/k
    assert float((wgt.sum(2) - 1).abs().max()) < 1e-5
    assert int(idx.min()) >= 0 and int(idx.max()) < frames * hw


@pytest.mark.parametrize("T,h,w,K", [(1, 8, 10, 1), (3, 9, 13, 2), (5, 30, 54, 3), (7, 30, 54, 5), (2, 68, 120, 1)])
def test_memory_read_full_softmax_vs_oracle(T, h, w, K):
    """top_k=None (prop_net.py:99-102, the reference's "no top-k" configuration): one-pass softmax over all memory positions +
    readout on exact fp32 MFMA (csrc/memory_read_dense.hip) vs the oracle's materialised F.softmax(affinity) @ values in fp64 and
    fp32; fp32 rows and the SH32 activation outputs of the decoder path; ragged tiles (T*h*w not a multiple of 16), several
    memory segments per query tile, the 480p benchmark shape and the 1080p grid."""
    mk, mv, qk = _mem_case(T, h, w, K, seed=900 + T * 10 + K, scale=1.5)
    keys = mk.permute(0, 2, 3, 4, 1).reshape(K, T * h * w, 128).contiguous().to(DEV)
    vals = mv.permute(0, 2, 3, 4, 1).reshape(K, T * h * w, 512).contiguous().to(DEV)
    q = qk.permute(0, 2, 3, 1).reshape(h * w, 128).contiguous().to(DEV)
    got = ops.memory_read(keys, vals, q, None)
    ref64 = torch.cat([O.memory_read(mk[o:o + 1].double(), mv[o:o + 1].double(), qk.double(), None) for o in range(K)], 0)
    ref32 = torch.cat([O.memory_read(mk[o:o + 1], mv[o:o + 1], qk, None) for o in range(K)], 0)
    g = got.cpu().view(K, h, w, 512).permute(0, 3, 1, 2)
    e64, r64 = float((g.double() - ref64).abs().max()), float((ref32.double() - ref64).abs().max())
    print(f"full softmax T={T} {h}x{w} K={K}: max|d| vs fp64 {e64:.2e} (torch fp32 vs fp64: {r64:.2e}), readout range {float(ref64.abs().max()):.2f}")
    assert e64 < 2e-5 * max(1.0, float(ref64.abs().max()))
    raw, rel = ops.memory_read_acts(keys, vals, q, None, h, w, tag="test.dense")
    assert float((ops.to_f32(raw) - got.view(K, h, w, 512)).abs().max()) < 1e-5 * max(1.0, float(got.abs().max()))
    assert float((ops.to_f32(rel) - got.view(K, h, w, 512).clamp(min=0)).abs().max()) < 1e-5 * max(1.0, float(got.abs().max()))
    m4 = torch.full((K, h * w, 1024), 7.0, device=DEV)
    ops.memory_read(keys, vals, q, None, out=m4[:, :, :512])                     # channel-slice destination
    assert torch.equal(m4[:, :, :512], got) and float((m4[:, :, 512:] - 7).abs().max()) == 0


@pytest.mark.parametrize("T,h,w,K,top_k", [(3, 9, 13, 2, 65), (5, 30, 54, 2, 128), (2, 30, 54, 1, 1000), (1, 8, 10, 1, 80)])
def test_memory_read_large_k_vs_oracle(T, h, w, K, top_k):
    """k beyond the streaming kernels (64 < k <= 1024; top_k == THW too): scores + radix-select path (mivos_memory_read_topk_any) vs
    the oracle - exact index sets on clear-margin queries, readout, weights summing to one."""
    mk, mv, qk = _mem_case(T, h, w, K, seed=T * 1000 + top_k, scale=1.3)
    got, idx, wgt = _run_mem(mk, mv, qk, top_k)
    for o in range(K):
        ref = O.memory_read(mk[o:o + 1].double(), mv[o:o + 1].double(), qk.double(), top_k)
        a = O.affinity(mk[o:o + 1].double(), qk.double())[0]
        vals, ridx = torch.topk(a, min(top_k + 1, a.shape[0]), dim=0)
        clear = (vals[top_k - 1] - vals[top_k]) > 1e-5 if a.shape[0] > top_k else torch.ones(a.shape[1], dtype=torch.bool)
        d = (got[o].double() - ref[0]).abs().amax(0).reshape(-1)
        assert float(d[clear].max()) < 2e-4 and int((~clear).sum()) <= max(2, clear.numel() // 50)
        same = (torch.sort(idx[o].long(), dim=1)[0] == torch.sort(ridx[:top_k].t(), dim=1)[0]).all(dim=1)
        assert bool(same[clear].all()) and float((wgt[o].sum(1) - 1).abs().max()) < 1e-5


def test_memory_read_sharp_scores_and_ties(mem_precision):
    # large-magnitude keys (softmax nearly one-hot) and duplicated memory rows (exact score ties)
    mk, mv, qk = _mem_case(2, 8, 10, 1, seed=9, scale=6.0)
    mk[:, :, 1] = mk[:, :, 0]                                   # frame 1 duplicates frame 0: every score ties
    got, idx, _ = _run_mem(mk, mv, qk, 20)
    mv2 = mv.clone()
    ref = O.memory_read(mk, mv2, qk, 20)
    # with exact ties torch.topk's choice is unspecified: compare through tie-invariant values
    mv_t = mv.clone()
    mv_t[:, :, 1] = mv_t[:, :, 0]
    got_t, _, _ = _run_mem(mk, mv_t, qk, 20)
    ref_t = O.memory_read(mk, mv_t, qk, 20)
    assert float((got_t - ref_t).abs().max()) < 2e-4
    assert int(idx.max()) < 160 and int(idx.min()) >= 0                 # indices stay in range


def test_split_keys_matches_the_cpu_definition_bitwise_and_bank_path():
    """mivos_memory_split_keys vs oracle/sh32.split_key_rows, through a strided bank-slot view; a read that is handed the
    split bank equals the read that converts the keys itself."""
    from oracle import sh32
    g = torch.Generator().manual_seed(21)
    bank = torch.randn(3, 5, 6, 9, 128, generator=g) * 4                      # [K, slots, h, w, 128]
    dev = bank.to(DEV)
    split = torch.zeros_like(dev)
    ops.split_keys(dev[:, 1:4], split[:, 1:4])
    ops.split_keys(dev[:, 4], split[:, 4])
    assert torch.equal(split[:, 1:5].cpu().view(torch.int32), sh32.split_key_rows(bank[:, 1:5]).view(torch.int32))
    assert float(split[:, 0].abs().max()) == 0
    keys, ks = dev[:, 1:5].reshape(3, 4 * 54, 128), split[:, 1:5].reshape(3, 4 * 54, 128)
    vals = torch.randn(3, 4 * 54, 512, generator=g).to(DEV)
    q = (torch.randn(54, 128, generator=g) * 4).to(DEV)
    old, ops.CONV_PRECISION = ops.CONV_PRECISION, "f16x3"
    try:
        assert torch.equal(ops.memory_read(keys, vals, q, 20, keys_split=ks), ops.memory_read(keys, vals, q, 20))
    finally:
        ops.CONV_PRECISION = old


def test_memory_read_topk_out_of_range_raises():
    mk, mv, qk = _mem_case(1, 4, 6, 1, seed=1)                  # 24 positions < top_k = 50
    with pytest.raises(RuntimeError, match="out of range"):
        _run_mem(mk, mv, qk, 50)


def test_memory_read_writes_into_channel_slice():
    mk, mv, qk = _mem_case(2, 8, 10, 2, seed=11)
    keys = mk.permute(0, 2, 3, 4, 1).reshape(2, 160, 128).contiguous().to(DEV)
    vals = mv.permute(0, 2, 3, 4, 1).reshape(2, 160, 512).contiguous().to(DEV)
    q = qk.permute(0, 2, 3, 1).reshape(80, 128).contiguous().to(DEV)
    m4 = torch.full((2, 80, 1024), 7.0, device=DEV)
    ops.memory_read(keys, vals, q, 20, out=m4[:, :, :512])
    dense = ops.memory_read(keys, vals, q, 20)
    assert torch.equal(m4[:, :, :512], dense) and float((m4[:, :, 512:] - 7).abs().max()) == 0


def test_attention_align(golden_dir):
    with np.load(os.path.join(golden_dir, "ops_small.npz")) as z:
        mk, qk, pos, neg, ref = (torch.from_numpy(z[k]) for k in ("at_mk", "at_qk", "at_pos", "at_neg", "at_out"))
    from mivos_amd.model.propagation.prop_net import PropagationNetwork
    net = PropagationNetwork()
    got = net.get_attention(mk.to(DEV), pos.to(DEV), neg.to(DEV), qk.to(DEV)).cpu()
    assert float((got - ref).abs().max()) < 1e-5
    # 480p-sized random case, 3 objects, against the oracle
    g = torch.Generator().manual_seed(12)
    mk, qk = torch.randn(3, 128, 1, 30, 54, generator=g), torch.randn(1, 128, 30, 54, generator=g)
    pos, neg = torch.rand(3, 1, 480, 864, generator=g), torch.rand(3, 1, 480, 864, generator=g)
    got = net.get_attention(mk.to(DEV), pos.to(DEV), neg.to(DEV), qk.to(DEV)).cpu()
    ref = torch.cat([O.get_attention(mk[i:i + 1], pos[i:i + 1], neg[i:i + 1], qk) for i in range(3)], 0)
    assert float((got - ref).abs().max()) < 1e-5

"""Parameter containers + compiled HIP execution plans for the STM building blocks.

Mirrors the *interface* of the reference's `model/propagation/modules.py` and
`mod_resnet.py` (same attribute / state_dict names, so released checkpoints load with
``load_state_dict``), but the modules here never run torch convolutions: ``compile()``
folds eval-mode BatchNorm into a per-channel scale/bias, re-lays the weights out as OHWI
and every forward is a chain of fused implicit-GEMM launches on NHWC tensors
(mivos_amd/csrc/conv_igemm.hip).
"""
import math

import torch
import torch.nn as nn

from ... import ops
from ...ops import ConvLayer


import os

STEM_FROM_PLANES = os.environ.get("MIVOS_STEM_PLANES", "1") != "0"     # tuning / A-B only: 0 = interleave + implicit-GEMM stem


class ConvParams(nn.Module):
    """Holds `weight` (+ `bias`) of an nn.Conv2d; no torch forward."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, dilation=1, init="he_normal"):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding, self.dilation = cin, cout, k, stride, padding, dilation
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        if init == "conv2d":
            # nn.Conv2d.reset_parameters: what a network TRAINED FROM SCRATCH starts from in the reference (FusionNet, model/fusion_net.py)
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
            if self.bias is not None:
                bound = 1.0 / math.sqrt(cin * k * k)
                nn.init.uniform_(self.bias, -bound, bound)
        else:
            nn.init.normal_(self.weight, 0.0, math.sqrt(2.0 / (cin * k * k)))

    def pack(self, bn=None, cin_pad=None):
        return ConvLayer.pack(self.weight, self.bias, None if bn is None else bn.tensors(), self.stride,
                              self.padding, cin_pad=cin_pad, eps=1e-5 if bn is None else bn.eps, dilation=self.dilation)

    def forward(self, *a, **k):
        raise RuntimeError("ConvParams is a parameter container; use the owning network's methods")


class BatchNormParams(nn.Module):
    """Eval-mode BatchNorm2d statistics/affine (state_dict compatible with nn.BatchNorm2d)."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def tensors(self):
        return self.weight, self.bias, self.running_mean, self.running_var


class Bottleneck(nn.Module):
    """ResNet-50 v1.5 bottleneck (mod_resnet.py:76-112; torchvision): 1x1 -> 3x3(stride) -> 1x1."""
    expansion = 4

    def __init__(self, cin, width, stride, bias, downsample, dilation=1):
        super().__init__()
        self.conv1 = ConvParams(cin, width, 1, bias=bias)
        self.bn1 = BatchNormParams(width)
        self.conv2 = ConvParams(width, width, 3, stride=stride, padding=dilation, bias=bias, dilation=dilation)
        self.bn2 = BatchNormParams(width)
        self.conv3 = ConvParams(width, width * 4, 1, bias=bias)
        self.bn3 = BatchNormParams(width * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(ConvParams(cin, width * 4, 1, stride=stride, bias=bias),
                                            BatchNormParams(width * 4))

    def compile(self):
        ds = None if self.downsample is None else self.downsample[0].pack(self.downsample[1])
        return (self.conv1.pack(self.bn1), self.conv2.pack(self.bn2), self.conv3.pack(self.bn3), ds)


def run_bottleneck(plan, x, tag=None):
    """x: fp32 NHWC tensor (register-staged kernels) or ops.Act (LDS-DMA kernels: intermediates live in scratch Acts, the
    block output in the scratch Act `tag`, or in fresh storage when tag is None)."""
    c1, c2, c3, ds = plan
    if isinstance(x, ops.Act):
        t = ops.conv(x, c1, relu_out=True, out_act=True, tag="bneck.1")
        t = ops.conv(t, c2, relu_out=True, out_act=True, tag="bneck.2")
        idt = x if ds is None else ops.conv(x, ds, out_act=True, tag="bneck.ds")
        return ops.conv(t, c3, res=idt, relu_out=True, out_act=True, tag=tag)
    t = ops.conv(x, c1, relu_out=True)
    t = ops.conv(t, c2, relu_out=True)
    idt = x if ds is None else ops.conv(x, ds)
    return ops.conv(t, c3, res=idt, relu_out=True)      # relu(bn3(conv3) + identity)


def _stage(cin, width, depth, stride, bias):
    blocks = [Bottleneck(cin, width, stride, bias, downsample=True)]
    blocks += [Bottleneck(width * 4, width, 1, bias, downsample=False) for _ in range(depth - 1)]
    return nn.Sequential(*blocks)


class _Trunk(nn.Module):
    """ResNet-50 stem + stages 1-3.  `first` names stage 1 ('layer1' or 'res2', modules.py:48,76)."""

    def __init__(self, in_ch, bias, first):
        super().__init__()
        self._first = first
        self.conv1 = ConvParams(in_ch, 64, 7, stride=2, padding=3, bias=bias)
        self.bn1 = BatchNormParams(64)
        setattr(self, first, _stage(64, 64, 3, 1, bias))
        self.layer2 = _stage(256, 128, 4, 2, bias)
        self.layer3 = _stage(512, 256, 6, 2, bias)

    def compile(self):
        cin_pad = 4 if self.conv1.cin <= 4 else 8          # channel axis padded to a power of two
        stages = [[b.compile() for b in getattr(self, n)] for n in (self._first, "layer2", "layer3")]
        return (self.conv1.pack(self.bn1, cin_pad=cin_pad), stages)


def run_trunk_planes(plan, planes, n, H, W, keep=True):
    """run_trunk on PLANAR input channels [(tensor, batch_stride), ...] (the frame's three planes, plus mask / others for the
    mask encoder: modules.py:54 cat([f, m, o])): with the f16x3 back-end the stem gathers them itself (mivos_stem7x7s2_planes),
    otherwise they are interleaved into the NHWC tensor run_trunk takes."""
    stem, stages = plan
    if ops.act_path() and STEM_FROM_PLANES:
        return run_trunk(plan, None, keep, stem_out=ops.stem_planes(planes, n, H, W, stem))
    return run_trunk(plan, ops.interleave(planes, n, H * W, stem.cin, stem.w.device).view(n, H, W, stem.cin), keep)


def run_trunk(plan, x, keep=True, stem_out=None):
    """x NHWC [N,H,W,4|8] -> (f16, f8, f4).  With the f16x3 back-end everything behind the stem runs on the LDS-DMA
    kernels and the features are ops.Act (SH32, zero-bordered); keep=False puts them into scratch storage too (the
    caller consumes them before the next trunk call).  stem_out: the stem's output when the caller ran it already."""
    stem, stages = plan
    x = stem_out if stem_out is not None else ops.conv(x, stem, relu_out=True)
    x = ops.maxpool3x3s2(x, act_tag="trunk.stem", as_act=True) if ops.act_path() else ops.maxpool3x3s2(x)   # SH32 from the pool itself
    feats = []
    for stage in stages:
        for i, blk in enumerate(stage):
            last = i == len(stage) - 1
            x = run_bottleneck(blk, x, tag=None if (last and keep) else ("bneck.out", i & 1))
        feats.append(x)
    return feats[2], feats[1], feats[0]


class MaskRGBEncoder(_Trunk):
    """modules.py:38-64: 5-channel (RGB + mask + others) ResNet-50 with biased convs."""

    def __init__(self):
        super().__init__(5, True, "layer1")


class RGBEncoder(_Trunk):
    """modules.py:67-89: torchvision ResNet-50 (bias-free convs); stage 1 is called `res2`."""

    def __init__(self):
        super().__init__(3, False, "res2")


class KeyValue(nn.Module):
    """modules.py:107-114.  Compiled as ONE 3x3 GEMM with 128+512 output channels whose epilogue
    writes the key and the value halves to two destinations (e.g. two memory-bank slots)."""

    def __init__(self, indim, keydim, valdim):
        super().__init__()
        self.key_proj = ConvParams(indim, keydim, 3, padding=1)
        self.val_proj = ConvParams(indim, valdim, 3, padding=1)

    def compile(self):
        return ConvLayer.fuse_outputs(self.key_proj.pack(), self.val_proj.pack())


class ResBlock(nn.Module):
    """modules.py:15-35 (pre-activation residual block, optional 3x3 conv on the skip)."""

    def __init__(self, indim, outdim=None):
        super().__init__()
        outdim = indim if outdim is None else outdim
        self.downsample = None if indim == outdim else ConvParams(indim, outdim, 3, padding=1)
        self.conv1 = ConvParams(indim, outdim, 3, padding=1)
        self.conv2 = ConvParams(outdim, outdim, 3, padding=1)

    def compile(self):
        return (self.conv1.pack(), self.conv2.pack(), None if self.downsample is None else self.downsample.pack())


def run_resblock(plan, x):
    """x fp32 NHWC -> fp32 NHWC: x(+ds) + conv2(relu(conv1(relu(x)))).  On the LDS-DMA path the two ReLUs move to the
    producers (the pack of x and the epilogue of conv1): a DMA-staged operand cannot be modified on load."""
    c1, c2, ds = plan
    if ops.act_path() and c1.cin % 32 == 0:
        r = ops.conv(ops.to_act(x, relu=True, tag="resblock.in"), c1, relu_out=True, out_act=True, tag="resblock.mid")
        skip = x if ds is None else ops.conv(ops.to_act(x, tag="resblock.raw"), ds)
        return ops.conv(r, c2, res=skip)
    r = ops.conv(x, c1, relu_in=True)
    skip = x if ds is None else ops.conv(x, ds)
    return ops.conv(r, c2, relu_in=True, res=skip)


def run_resblock_acts(plan, raw, rel, res1=None, res_skip=None):
    """The same block on pre-split operands: `raw` / `rel` are SH32 Acts of x and relu(x) written by x's producer.
    res1 / res_skip: partial sums (fp32, batch 1, broadcast) added to conv1 / to the skip convolution - the contribution of
    input channels that are identical for every object and were convolved once (Decoder.compress, see prop_net.segment)."""
    c1, c2, ds = plan
    # (Round 6 measured the skip convolution / the stages' projection shortcuts on a side stream beside conv1: same box, one clip in flight 201.9 vs 201.9
    # frames/s, two clips 187 vs 223 - the extra stream's event traffic costs more than the overlap gives; profiles/r06d_branch_stream_ab.txt.  Not kept.)
    r = ops.conv(rel, c1, relu_out=True, res=res1, out_act=True, tag="resblock.mid")
    skip = raw if ds is None else ops.conv(raw, ds, res=res_skip)
    return ops.conv(r, c2, res=skip)


class UpsampleBlock(nn.Module):
    """modules.py:92-104."""

    def __init__(self, skip_c, up_c, out_c, scale_factor=2):
        super().__init__()
        assert scale_factor == 2
        self.skip_conv1 = ConvParams(skip_c, up_c, 3, padding=1)
        self.skip_conv2 = ResBlock(up_c, up_c)
        self.out_conv = ResBlock(up_c, out_c)

    def compile(self):
        return (self.skip_conv1.pack(), self.skip_conv2.compile(), self.out_conv.compile())


def run_skip_branch(plan, skip_f):
    """The object-independent half of UpsampleBlock.forward: skip_conv2(skip_conv1(skip_f))."""
    sc1, sc2, _ = plan
    return run_resblock(sc2, ops.conv(skip_f, sc1))


def run_up_branch(plan, skip_feat, up_f, tag="up"):
    """out_conv(skip_feat + bilinear_x2(up_f)); skip_feat is broadcast over the object batch.  On the LDS-DMA path the
    upsample kernel writes the sum pre-split, as x and as relu(x): no conversion pass before the ResBlock."""
    if ops.act_path() and up_f.shape[3] % 32 == 0:
        raw, rel = ops.upsample2x_add_acts(skip_feat, up_f, tag)
        return run_resblock_acts(plan[2], raw, rel)
    return run_resblock(plan[2], ops.upsample2x_add(skip_feat, up_f))

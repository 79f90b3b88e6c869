"""Difference-aware FusionNet for MI355X — interface of the reference's `model/fusion_net.py:8-50`
(same constructor, forward signature and 12-key state_dict), run as fused NHWC implicit-GEMM
convolutions (bias + residual + ReLU in the GEMM epilogue, 32->1 head as a dot-product kernel).
"""
import torch
import torch.nn as nn

from .. import ops
from .._lib import MivosHipError
from .plan_cache import PlanCache
from .propagation.modules import ConvParams


import os

ONE_CALL = os.environ.get("MIVOS_FUSION_ONE_CALL", "1") != "0"     # tuning / A-B only: 0 = the layer-by-layer path (six launches + tap sum)


class FusionNet(PlanCache):
    def __init__(self):
        super().__init__()
        conv = lambda cin, cout: ConvParams(cin, cout, 3, padding=1, init="conv2d")       # trained from scratch: nn.Conv2d's default init
        self.conv1 = nn.Sequential(conv(9, 32), nn.ReLU())
        self.conv2 = nn.Sequential(conv(32, 32), nn.ReLU(), conv(32, 32))
        self.conv3 = nn.Sequential(conv(32, 32), nn.ReLU(), conv(32, 32))
        self.relu = nn.ReLU()
        self.final_conv = conv(32, 1)

    def plan(self):
        if self._plan is None:
            if self.final_conv.weight.device.type != "cuda":
                raise MivosHipError("FusionNet must live on an MI355X; mivos_amd has no CPU execution path")
            with torch.no_grad():
                self._plan = (self.conv1[0].pack(cin_pad=16), self.conv2[0].pack(), self.conv2[2].pack(),
                              self.conv3[0].pack(), self.conv3[2].pack(), self.final_conv.pack())
                self._stamp_plan()
                ops.publish_constants()          # read by launches on every stream from here on
        return self._plan

    def run(self, x, layered=False):
        """x NHWC [B,H,W,16] (9 real channels: im, seg1, seg2, attn(2), time(2)) -> logits [B,H,W,1].
        Default (f16x3 back-end): one mivos_fusion_net_forward call = conv1 + two fused residual-block launches + the fp32
        head (csrc/fusion_net.hip).  layered=True (and the exact-fp32 verification mode) issues the six convolutions one by
        one through ops.conv - the reference's structure, kept as the checker of the fused kernels."""
        c1, c2a, c2b, c3a, c3b, fin = self.plan()
        if ops.CONV_PRECISION == "f16x3" and ONE_CALL and not layered:
            if ops.PROFILE is None:
                return ops.fusion_net_forward(x, (c1, c2a, c2b, c3a, c3b), fin)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            out = ops.fusion_net_forward(x, (c1, c2a, c2b, c3a, c3b), fin)
            ev1.record()
            px = x.shape[0] * x.shape[1] * x.shape[2]
            # bench.py: the whole network as one sample (variant 30); FLOP = 2 * MACs of the six convolutions (conv1 on its 16 padded channels)
            ops.PROFILE.append((30, 2.0 * px * (9 * 16 * 32 + 4 * 9 * 32 * 32 + 9 * 32), ev0, ev1, ("fusion_net", px, 4.0 * px * 17)))
            return out
        x = ops.conv(x, c1, relu_out=True)
        r = ops.conv(x, c2a, relu_out=True)
        x = ops.conv(r, c2b, res=x, relu_out=True)        # relu(x + conv2(x))   fusion_net.py:42-43
        r = ops.conv(x, c3a, relu_out=True)
        x = ops.conv(r, c3b, res=x, relu_out=True)        # relu(x + conv3(x))   fusion_net.py:45-46
        return ops.conv(x, fin)

    @staticmethod
    def _planes(im, seg1, seg2, attn, time_pair):
        """The nine input channels of fusion_net.py:38 as planes.  Each of im/seg1/seg2/attn is (tensor, batch_stride_in_elements);
        time_pair is (nc, nr) python floats."""
        (im_t, im_s), (s1_t, s1_s), (s2_t, s2_s), (at_t, at_s) = im, seg1, seg2, attn
        H, W = im_t.shape[-2:]
        P = H * W
        imf, atf = im_t.reshape(-1), at_t.reshape(-1)
        planes = [(imf[c * P:], im_s) for c in range(3)] + [(s1_t, s1_s), (s2_t, s2_s)]
        planes += [(atf[c * P:], at_s) for c in range(2)] + [(float(time_pair[0]), 0), (float(time_pair[1]), 0)]
        return planes, H, W

    def pack_inputs(self, im, seg1, seg2, attn, time_pair, batch):
        """Channel-concatenate the planar inputs into NHWC16 (the layer-by-layer path's input)."""
        planes, H, W = self._planes(im, seg1, seg2, attn, time_pair)
        return ops.interleave(planes, batch, H * W, 16, im[0].device).view(batch, H, W, 16)

    def run_planes(self, im, seg1, seg2, attn, time_pair, batch):
        """run() on planar inputs (arguments of pack_inputs): with the f16x3 back-end conv1 gathers the planes itself
        (mivos_fusion_conv1_planes) and the 16-channel concatenation is never materialised."""
        if not (ops.CONV_PRECISION == "f16x3" and ONE_CALL):
            return self.run(self.pack_inputs(im, seg1, seg2, attn, time_pair, batch))
        c1, c2a, c2b, c3a, c3b, fin = self.plan()
        planes, H, W = self._planes(im, seg1, seg2, attn, time_pair)
        ev = None
        if ops.PROFILE is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        out = ops.fusion_net_forward(None, (c1, c2a, c2b, c3a, c3b), fin, planes=planes, shape=(batch, H, W))
        if ev is not None:
            ev[1].record()
            px = batch * H * W
            ops.PROFILE.append((30, 2.0 * px * (9 * 16 * 32 + 4 * 9 * 32 * 32 + 9 * 32), ev[0], ev[1], ("fusion_net", px, 4.0 * px * 10)))
        return out

    def forward(self, im, seg1, seg2, attn, time):
        B, _, H, W = im.shape
        P = H * W
        with ops.on_device(im):
            im, seg1, seg2, attn = (t.contiguous().float() for t in (im, seg1, seg2, attn))
            outs = []
            tl = time.detach().float().cpu().tolist()
            for b in range(B):   # per-sample constant time planes (B == 1 on the inference path)
                outs.append(self.run_planes((im[b], 0), (seg1[b], 0), (seg2[b], 0), (attn[b], 0), tl[b], 1))
            y = outs[0] if B == 1 else torch.cat(outs, 0)
            return y.permute(0, 3, 1, 2)

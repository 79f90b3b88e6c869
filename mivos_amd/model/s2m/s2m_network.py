"""Scribble-to-mask network for MI355X — interface of the reference's `model/s2m/s2m_network.py`
(``deeplabv3plus_resnet50(num_classes=1, output_stride=16)``: DeepLabV3+ on a 6-channel ResNet-50, `_deeplab.py:30-164`,
`s2m_resnet.py`), same state_dict names, executed on the HIP convolution engine.

It is the step right before the propagation path in the interactive benchmark (`davis_processor.py:38-70`: one forward per
object per interaction).  Execution plan:
  * stem + stages 1-3: the same LDS-DMA bottleneck chain as the STM encoders (SH32 activations);
  * stage 4 (stride replaced by dilation 2) and the three atrous 3x3 branches of ASPP (dilation 6 / 12 / 18): the register-staged
    f16x3 kernels with the `dilation` field of mivos_conv2d_fused;
  * ASPP: the four spatial branches write channel slices of one [h,w,1024] buffer (the concatenation is a stride); the image-pooling
    branch is constant over the image, so its share of the 1x1 projection is folded into that convolution's bias;
  * head: low-level projection and the x4 upsampled ASPP output land in channel slices of one 320-channel buffer
    (48 + 16 zero + 256; the classifier's weights are re-indexed accordingly), 3x3 classifier on the LDS-DMA kernels.
"""
import torch
import torch.nn as nn

from ... import ops
from ..._lib import MivosHipError
from ..plan_cache import PlanCache
from ..propagation.modules import BatchNormParams, Bottleneck, ConvParams, run_bottleneck


def _stage(cin, width, depth, stride, dilation_first, dilation_rest):
    blocks = [Bottleneck(cin, width, stride, False, downsample=True, dilation=dilation_first)]
    blocks += [Bottleneck(width * 4, width, 1, False, downsample=False, dilation=dilation_rest) for _ in range(depth - 1)]
    return nn.Sequential(*blocks)


class _Backbone(nn.Module):
    """s2m_resnet.ResNet (6 input channels: RGB, current mask, positive / negative scribbles) up to layer4, with
    replace_stride_with_dilation = [False, False, True] (output stride 16)."""

    def __init__(self):
        super().__init__()
        self.conv1 = ConvParams(6, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNormParams(64)
        self.relu, self.maxpool = nn.Identity(), nn.Identity()          # parameter-free slots of the reference's ModuleDict
        self.layer1 = _stage(64, 64, 3, 1, 1, 1)
        self.layer2 = _stage(256, 128, 4, 2, 1, 1)
        self.layer3 = _stage(512, 256, 6, 2, 1, 1)
        self.layer4 = _stage(1024, 512, 3, 1, 1, 2)                     # _make_layer(dilate=True): first block keeps dilation 1


def _conv_bn(cin, cout, k, dilation=1):
    return nn.Sequential(ConvParams(cin, cout, k, padding=dilation if k == 3 else 0, bias=False, dilation=dilation), BatchNormParams(cout), nn.Identity())


class _ASPP(nn.Module):
    def __init__(self, cin, rates):
        super().__init__()
        convs = [_conv_bn(cin, 256, 1)] + [_conv_bn(cin, 256, 3, r) for r in rates]
        convs.append(nn.Sequential(nn.Identity(), ConvParams(cin, 256, 1, bias=False), BatchNormParams(256), nn.Identity()))   # ASPPPooling
        self.convs = nn.ModuleList(convs)
        self.project = nn.Sequential(ConvParams(5 * 256, 256, 1, bias=False), BatchNormParams(256), nn.Identity(), nn.Identity())


class _HeadV3Plus(nn.Module):
    def __init__(self, cin, low_c, num_classes, rates):
        super().__init__()
        self.project = nn.Sequential(ConvParams(low_c, 48, 1, bias=False), BatchNormParams(48), nn.Identity())
        self.aspp = _ASPP(cin, rates)
        self.classifier = nn.Sequential(ConvParams(304, 256, 3, padding=1, bias=False), BatchNormParams(256), nn.Identity(),
                                        ConvParams(256, num_classes, 1))


class S2M(PlanCache):
    def __init__(self, num_classes=1):
        super().__init__()
        if num_classes != 1:
            raise MivosHipError("S2M: the scribble-to-mask network has one output class")
        self.backbone = _Backbone()
        self.classifier = _HeadV3Plus(2048, 256, num_classes, (6, 12, 18))

    def plan(self):
        if self._plan is None:
            if self.backbone.conv1.weight.device.type != "cuda":
                raise MivosHipError("S2M must live on an MI355X; mivos_amd has no CPU execution path")
            with torch.no_grad():
                b, c = self.backbone, self.classifier
                stages = [[blk.compile() for blk in getattr(b, n)] for n in ("layer1", "layer2", "layer3", "layer4")]
                aspp = [c.aspp.convs[i][0].pack(c.aspp.convs[i][1]) for i in range(4)]
                pool = c.aspp.convs[4][1].pack(c.aspp.convs[4][2])
                proj = c.aspp.project[0].pack(c.aspp.project[1])                     # 1x1, 1280 -> 256, BN folded into scale / bias
                # the projection over the four spatial branches: its BN shift travels with the pooling branch's term (proj_pool)
                proj_main = ops.ConvLayer(proj.w[..., :1024].contiguous(), proj.scale, None, 1, 0)
                proj_pool = ops.ConvLayer(proj.w[..., 1024:].contiguous(), proj.scale, proj.bias, 1, 0)
                # low-level projection: 48 output channels padded to 64 (zero rows, zero shift: relu(0) = 0)
                lp = c.project[0].pack(c.project[1])
                pad = lambda t, fill: torch.cat([t, t.new_full((16,) + t.shape[1:], fill)], 0)
                low = ops.ConvLayer(pad(lp.w, 0.0), pad(lp.scale, 1.0).contiguous(), pad(lp.bias, 0.0).contiguous(), 1, 0)
                # classifier conv over cat([low (48), aspp (256)]): input channels re-indexed to [low 48 | 16 zeros | aspp 256]
                cl = c.classifier[0].pack(c.classifier[1])
                w = cl.w.new_zeros((256, 3, 3, 320))
                w[..., :48], w[..., 64:] = cl.w[..., :48], cl.w[..., 48:]
                cls0 = ops.ConvLayer(w, cl.scale, cl.bias, 1, 1)
                self._plan = dict(stem=b.conv1.pack(b.bn1, cin_pad=8), stages=stages, aspp=aspp, pool=pool, proj_main_nobias=proj_main,
                                  proj_pool=proj_pool, low=low, cls0=cls0, cls1=c.classifier[3].pack())
                self._stamp_plan()
        return self._plan

    def forward(self, x):
        """x [N,6,H,W] (RGB, mask, positive scribble, negative scribble; H, W multiples of 16) -> logits [N,1,H,W]."""
        N, C, H, W = x.shape
        if C != 6 or H % 16 or W % 16:
            raise MivosHipError(f"S2M.forward: expected [N,6,H,W] with H, W multiples of 16, got {tuple(x.shape)}")
        with ops.on_device(x), torch.no_grad():
            p = self.plan()
            x = x.contiguous().float()
            cap = ops.max_act_batch(H // 4, W // 4, 256)                                # 32-bit offsets of the LDS-DMA kernels
            if N > cap:
                return torch.cat([self._forward_batch(p, x[n:n + cap], H, W) for n in range(0, N, cap)], 0)
            return self._forward_batch(p, x, H, W)

    def _forward_batch(self, p, x, H, W):
        """All N samples in one chain of launches (DAVISProcessor hands over one sample per object of the interaction,
        davis_processor.py:52-70).  The image-pooling branch of ASPP is constant over an image: its share of the 1x1 projection
        enters that convolution as a per-image vector, passed as a residual whose row / pixel strides are zero."""
        N, P = x.shape[0], H * W
        flat = x.reshape(-1)
        xin = ops.interleave([(flat[c * P:], 6 * P) for c in range(6)], N, P, 8, x.device).view(N, H, W, 8)
        y = ops.conv(xin, p["stem"], relu_out=True)
        act = ops.act_path()
        y = ops.maxpool3x3s2(y, act_tag="s2m.stem", as_act=True) if act else ops.maxpool3x3s2(y)
        low_level = None
        for si, stage in enumerate(p["stages"]):
            if si == 3 and act:
                y = ops.to_f32(y)                                          # stage 4 is dilated: register-staged kernels, fp32 tensors
            for bi, blk in enumerate(stage):
                last = bi == len(stage) - 1
                y = run_bottleneck(blk, y, tag=None if (last and si == 0) else ("s2m.bneck", bi & 1))
            if si == 0:
                low_level = y
        h, w = H // 16, W // 16
        cat = torch.empty((N, h, w, 1024), dtype=torch.float32, device=x.device)
        for i, L in enumerate(p["aspp"]):
            ops.conv(y, L, relu_out=True, out=cat[..., 256 * i:256 * (i + 1)])
        pooled = ops.conv(ops.global_avgpool(y), p["pool"], relu_out=True)             # [N,1,1,256], constant over each image
        pool_term = ops.conv(pooled, p["proj_pool"])                                    # scale * (W_pool . pooled) + BN shift, [N,1,1,256]
        aspp_out = ops.conv(cat, p["proj_main_nobias"], relu_out=True, res=pool_term.expand(N, h, w, 256))   # Dropout(0.1): identity in eval
        cat2 = torch.zeros((N, H // 4, W // 4, 320), dtype=torch.float32, device=x.device)
        ops.conv(low_level, p["low"], relu_out=True, out=cat2[..., :64])
        ops.resize_bilinear_nhwc(aspp_out, H // 4, W // 4, out=cat2[..., 64:])
        z = ops.conv(ops.to_act(cat2, tag="s2m.cat"), p["cls0"], relu_out=True) if act else self._cls0_f32(p, cat2)
        lo = ops.conv(z, p["cls1"])                                                     # [N,H/4,W/4,1]
        return ops.resize_bilinear(lo.view(N, H // 4, W // 4), H, W).view(N, 1, H, W)

    @staticmethod
    def _cls0_f32(p, cat2):
        # exact-fp32 verification mode: the implicit-GEMM kernels need a power-of-two channel count
        n, h, w, _ = cat2.shape
        x = torch.zeros((n, h, w, 512), dtype=torch.float32, device=cat2.device)
        x[..., :320] = cat2
        L = p["cls0"]
        w512 = L.w.new_zeros((256, 3, 3, 512))
        w512[..., :320] = L.w
        return ops.conv(x, ops.ConvLayer(w512, L.scale, L.bias, 1, 1), relu_out=True)


def deeplabv3plus_resnet50(num_classes=1, output_stride=16, pretrained_backbone=False):
    """Same factory name / arguments as the reference (`s2m_network.py:56-65`); output stride 16 only."""
    if output_stride != 16:
        raise MivosHipError("S2M: only output_stride=16 (the configuration MiVOS ships) is implemented")
    return S2M(num_classes)

"""AttentionReadNetwork — training-time twin of PropagationNetwork.get_attention
(reference `model/attn_network.py:30-80`): frozen encoders, T=1 attention, positive/negative
difference maps for two objects.  Same state_dict names; same HIP kernels as the inference path."""
import torch
import torch.nn as nn

from .. import ops
from .._lib import MivosHipError
from .plan_cache import PlanCache
from .propagation.modules import KeyValue, MaskRGBEncoder, RGBEncoder, run_trunk


class AttentionMemory(nn.Module):
    def __init__(self, k=50):
        super().__init__()
        self.k = k

    def forward(self, mk, qk):
        """attn_network.py:17-28: mk, qk [B,128,H,W] -> W [B, HW, HW] (softmax over the memory positions; every
        sample has its own query map).  AttentionReadNetwork.forward itself uses the fused kernel and never builds W."""
        B, CK, H, W = mk.shape
        with ops.on_device(mk):
            keys = mk.permute(0, 2, 3, 1).reshape(B, H * W, CK)
            q = qk.permute(0, 2, 3, 1).reshape(B, H * W, CK)
            return ops.attention_weights(keys, q)


class AttentionReadNetwork(PlanCache):
    def __init__(self):
        super().__init__()
        self.mask_rgb_encoder = MaskRGBEncoder()
        self.rgb_encoder = RGBEncoder()
        self.kv_m_f16 = KeyValue(1024, keydim=128, valdim=512)
        self.kv_q_f16 = KeyValue(1024, keydim=128, valdim=512)
        self.memory = AttentionMemory()
        for p in self.parameters():
            p.requires_grad = False

    def plan(self):
        if self._plan is None:
            if self.kv_q_f16.key_proj.weight.device.type != "cuda":
                raise MivosHipError("AttentionReadNetwork must live on an MI355X")
            with torch.no_grad():
                self._plan = dict(menc=self.mask_rgb_encoder.compile(), qenc=self.rgb_encoder.compile(),
                                  kv_m=self.kv_m_f16.compile(), kv_q=self.kv_q_f16.compile())
                self._stamp_plan()
        return self._plan

    def _mem_keys(self, image, mask, other):
        p = self.plan()
        B, _, H, W = image.shape
        P = H * W
        imf = image.contiguous().reshape(-1)
        planes = [(imf[c * P:], 3 * P) for c in range(3)] + [(mask.contiguous(), P), (other.contiguous(), P)]
        x = ops.interleave(planes, B, P, 8, image.device).view(B, H, W, 8)
        f16, _, _ = run_trunk(p["menc"], x)
        k16, _ = ops.conv(f16, p["kv_m"])
        return k16

    def forward(self, image, mask11, mask21, mask12, mask22, query_image):
        b, _, h, w = mask11.shape
        nh, nw = h // 16, w // 16
        p = self.plan()
        with torch.no_grad(), ops.on_device(image):
            pos1, neg1 = ops.mask_diff(mask21, mask11)        # clamp(m21 - m11, 0, 1), clamp(m11 - m21, 0, 1)
            pos2, neg2 = ops.mask_diff(mask22, mask12)
            k1 = self._mem_keys(image, mask21, mask22)
            k2 = self._mem_keys(image, mask22, mask21)
            P = h * w
            qf = query_image.contiguous().reshape(-1)
            xq = ops.interleave([(qf[c * P:], 3 * P) for c in range(3)], b, P, 4, image.device).view(b, h, w, 4)
            qf16, _, _ = run_trunk(p["qenc"], xq)
            qk16, _ = ops.conv(qf16, p["kv_q"])
            outs = []
            for keys, pos, neg in ((k1, pos1, neg1), (k2, pos2, neg2)):
                p16 = ops.area_pool16(pos.reshape(b, h, w)).view(b, nh * nw)
                n16 = ops.area_pool16(neg.reshape(b, h, w)).view(b, nh * nw)
                maps = []
                for i in range(b):   # each sample has its own query key map
                    low = ops.attention_align(keys[i:i + 1].reshape(1, nh * nw, 128), qk16[i].reshape(nh * nw, 128),
                                              p16[i:i + 1], n16[i:i + 1])
                    maps.append(ops.resize_bilinear(low.view(2, nh, nw), h, w))
                outs.append(torch.stack(maps, 0))
        return outs[0], outs[1]

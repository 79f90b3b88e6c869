"""PropagationNetwork for MI355X — same public surface as the reference's
`model/propagation/prop_net.py:131-200` (memorize / get_query_values / segment_with_query /
get_W / get_attention, ``top_k`` constructor argument, 597-key state_dict), executed by
hand-written HIP kernels through libmivos_hip.so.

Layout.  Public methods take and return tensors with the reference's *logical* shapes
(NCHW, keys ``[K,128,T,h,w]`` ...).  Physically everything is channels-last: a returned
``[N,C,H,W]`` tensor is a zero-copy permuted view of an NHWC buffer, and memory keys/values are
rows of 128/512 floats per memory position (``[K,T,h,w,C]``), which is what the affinity MFMA
tiles and the sparse value gather want.  Inputs in any other layout are converted on entry.

``InferenceCore`` uses the internal fast path (``encode_query`` / ``memorize_into`` /
``segment``) that also caches the object-independent decoder skip branches per frame.
"""
import torch
import torch.nn as nn

from ... import ops
from ..._lib import MivosHipError
from ..plan_cache import PlanCache
from .modules import (ConvParams, KeyValue, MaskRGBEncoder, ResBlock, RGBEncoder, UpsampleBlock,
                      run_resblock, run_resblock_acts, run_skip_branch, run_trunk_planes, run_up_branch)

CK, CV = 128, 512


class Decoder(nn.Module):
    """prop_net.py:14-31."""

    def __init__(self):
        super().__init__()
        self.compress = ResBlock(1024, 512)
        self.up_16_8 = UpsampleBlock(512, 512, 256)
        self.up_8_4 = UpsampleBlock(256, 256, 256)
        self.pred = ConvParams(256, 1, 3, padding=1)

    def compile(self):
        c1, c2, ds = self.compress.compile()
        # compress reads cat([memory readout (512, per object), v16 (512, the SAME for every object)]) (prop_net.py:178-179):
        # a convolution over concatenated channels is the sum of the convolutions over the parts, so the v16 half is
        # convolved once per frame (batch 1, cached with the query features) and added as a broadcast partial sum
        half = c1.cin // 2
        split = dict(c1m=c1.slice_cin(0, half, False), c1v=c1.slice_cin(half, c1.cin, True),
                     dsm=ds.slice_cin(0, half, False), dsv=ds.slice_cin(half, ds.cin, True), c2=c2)
        return dict(compress=(c1, c2, ds), compress_split=split, up_16_8=self.up_16_8.compile(),
                    up_8_4=self.up_8_4.compile(), pred=self.pred.pack())


class EvalMemoryReader(nn.Module):
    """prop_net.py:75-108 (km=None): fused affinity + top-k softmax + readout."""

    def __init__(self, top_k, km=None):
        super().__init__()
        if km is not None:
            raise NotImplementedError("kernelised memory (km) is not used by the reference's inference path")
        self.top_k, self.km = top_k, km

    def forward(self, mk, mv, qk):
        B, _, T, H, W = mk.shape
        keys = mk.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, CK)
        vals = mv.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, CV)
        q = qk.permute(0, 2, 3, 1).reshape(H * W, CK).contiguous()
        with ops.on_device(mk):
            out = ops.memory_read(keys, vals, q, self.top_k)            # [B, HW, 512]
        return out.view(B, H, W, CV).permute(0, 3, 1, 2)


class AttentionMemory(nn.Module):
    """prop_net.py:110-129.  The [HW x HW] softmax matrix is only materialised if somebody asks
    for it (get_W); get_attention() uses the fused kernel that never writes it."""

    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, mk, qk):
        """mk [B,128,T,H,W], qk [B or 1,128,H,W] -> W [B, T*H*W, H*W], softmax over the memory axis."""
        B, _, T, H, W = mk.shape
        with ops.on_device(mk):
            keys = mk.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, CK)
            q = _nhwc(qk).reshape(qk.shape[0], H * W, CK)
            return ops.attention_weights(keys, q[0] if qk.shape[0] == 1 else q)


class QueryFeatures:
    """Per-frame query features, NHWC.  s8 / s4 (decoder skip branches, object independent) are
    filled lazily by PropagationNetwork._skip()."""
    __slots__ = ("f16", "f8", "f4", "k16", "v16", "s8", "s4", "c1v", "dsv")

    def __init__(self, f16, f8, f4, k16, v16):
        self.f16, self.f8, self.f4, self.k16, self.v16 = f16, f8, f4, k16, v16
        self.s8 = self.s4 = None
        self.c1v = self.dsv = None        # object-independent partial sums of Decoder.compress (the v16 half of its input)

    def as_reference_tuple(self):
        dense = [ops.to_f32(t) if isinstance(t, ops.Act) else t for t in (self.f16, self.f8, self.f4, self.k16, self.v16)]
        return tuple(t.permute(0, 3, 1, 2) for t in dense)


def _nhwc(t):
    """logical NCHW tensor -> NHWC tensor with unit channel stride (zero-copy when already channels-last)."""
    x = t.permute(0, 2, 3, 1)
    return x if x.is_contiguous() else x.contiguous()


MAX_TOP_K = 1024    # k <= 64 (the reference default is 50, the DAVIS single-object configuration uses 20) runs on the streaming
                    # select kernels of csrc/memory_read.hip, 64 < k <= 1024 on the scores + radix-select path of
                    # csrc/memory_read_dense.hip (ablations), top_k=None on the full-softmax kernel


class PropagationNetwork(PlanCache):
    def __init__(self, top_k=50):
        super().__init__()
        if top_k is not None and not (1 <= int(top_k) <= MAX_TOP_K):
            raise MivosHipError(f"PropagationNetwork(top_k={top_k}): the MI355X memory-read kernels support 1 <= top_k <= {MAX_TOP_K} "
                                f"(reference default 50), or top_k=None = softmax over the whole bank (prop_net.py:99-102)")
        self.mask_rgb_encoder = MaskRGBEncoder()
        self.rgb_encoder = RGBEncoder()
        self.kv_m_f16 = KeyValue(1024, keydim=CK, valdim=CV)
        self.kv_q_f16 = KeyValue(1024, keydim=CK, valdim=CV)
        self.memory = EvalMemoryReader(top_k, km=None)
        self.attn_memory = AttentionMemory(top_k)
        self.decoder = Decoder()

    # ---- compiled plan (packed weights; bookkeeping in model/plan_cache.py) -------------------
    def plan(self):
        if self._plan is None:
            dev = self.kv_q_f16.key_proj.weight.device
            if dev.type != "cuda":
                raise MivosHipError("PropagationNetwork must live on an MI355X (call .cuda() / .to('cuda:0')); "
                                    "mivos_amd has no CPU execution path")
            with torch.no_grad():
                self._plan = dict(menc=self.mask_rgb_encoder.compile(), qenc=self.rgb_encoder.compile(),
                                  kv_m=self.kv_m_f16.compile(), kv_q=self.kv_q_f16.compile(),
                                  dec=self.decoder.compile())
                self._stamp_plan()
                ops.publish_constants()          # read by launches on every stream from here on
        return self._plan

    # ---- internal fast path (NHWC) ------------------------------------------------------
    def encode_query(self, frame):
        """frame [1,3,H,W] (planar, normalised) -> QueryFeatures."""
        p = self.plan()
        _, _, H, W = frame.shape
        frame = frame.contiguous()
        f16, f8, f4 = run_trunk_planes(p["qenc"], [(frame[0, c], 0) for c in range(3)], 1, H, W)
        k16, v16 = ops.conv(f16, p["kv_q"])
        return QueryFeatures(f16, f8, f4, k16, v16)

    def encode_query_batch(self, frames, with_skip=True):
        """frames [B,3,H,W] -> B QueryFeatures (views into batched buffers).  Query features do not
        depend on the propagation state, so InferenceCore encodes the next few frames of a pass together:
        the batch-1 ResNet layers (30x54 = 1620 pixels) fill 10-40 % of the chip, B frames fill B times
        more and share one weight read.  `with_skip` also runs the decoder's object-independent skip
        branches for the whole batch."""
        p = self.plan()
        B, _, H, W = frames.shape
        cap = ops.max_act_batch(H // 4, W // 4, 256)          # 32-bit offsets of the LDS-DMA kernels (ops.ACT_BYTES_LIMIT)
        if B > cap:
            return [q for i in range(0, B, cap) for q in self.encode_query_batch(frames[i:i + cap], with_skip)]
        frames = frames.contiguous()
        P = H * W
        flat = frames.reshape(-1)
        f16, f8, f4 = run_trunk_planes(p["qenc"], [(flat[c * P:], 3 * P) for c in range(3)], B, H, W)
        k16, v16 = ops.conv(f16, p["kv_q"])
        s8 = s4 = c1v = dsv = None
        if with_skip:
            s8 = run_skip_branch(p["dec"]["up_16_8"], f8)
            s4 = run_skip_branch(p["dec"]["up_8_4"], f4)
            if ops.act_path():
                c1v, dsv = self._compress_v16(p["dec"]["compress_split"], v16)
        out = []
        for b in range(B):
            q = QueryFeatures(f16[b:b + 1], f8[b:b + 1], f4[b:b + 1], k16[b:b + 1], v16[b:b + 1])
            if with_skip:
                q.s8, q.s4 = s8[b:b + 1], s4[b:b + 1]
                if c1v is not None:
                    q.c1v, q.dsv = c1v[b:b + 1], dsv[b:b + 1]
            out.append(q)
        return out

    @staticmethod
    def _compress_v16(cs, v16):
        """The v16 half of Decoder.compress's two input convolutions (conv1 on relu(v16), the skip conv on v16), bias
        included: [B,h,w,512] fp32 each, identical for every object of the frame."""
        return (ops.conv(ops.to_act(v16, relu=True, tag="v16.relu"), cs["c1v"]), ops.conv(ops.to_act(v16, tag="v16.raw"), cs["dsv"]))

    def _skip(self, q):
        dec = self.plan()["dec"]
        if q.s8 is None:
            q.s8 = run_skip_branch(dec["up_16_8"], q.f8)
            q.s4 = run_skip_branch(dec["up_8_4"], q.f4)
        if q.c1v is None and ops.act_path():
            q.c1v, q.dsv = self._compress_v16(dec["compress_split"], q.v16)
        return q.s8, q.s4

    def memorize_into(self, frame, masks, key_out=None, val_out=None):
        """frame [1,3,H,W], masks [K,1,H,W] -> keys [K,h,w,128], values [K,h,w,512] (NHWC), written
        straight into key_out / val_out (e.g. a memory-bank slot view) when given."""
        p = self.plan()
        K, _, H, W = masks.shape
        frame, masks = frame.contiguous(), masks.contiguous().float()
        others = ops.mask_others(masks) if K > 1 else torch.zeros_like(masks)
        P = H * W
        cap = ops.max_act_batch(H // 4, W // 4, 256)
        if K > cap:                                             # object batch in chunks (tensors stay < 2 GB)
            if key_out is None:
                key_out = torch.empty((K, H // 16, W // 16, CK), dtype=torch.float32, device=frame.device)
                val_out = torch.empty((K, H // 16, W // 16, CV), dtype=torch.float32, device=frame.device)
            for i in range(0, K, cap):
                self._memorize_chunk(p, frame, masks[i:i + cap], others[i:i + cap], key_out[i:i + cap], val_out[i:i + cap])
            return key_out, val_out
        return self._memorize_chunk(p, frame, masks, others, key_out, val_out)

    def _memorize_chunk(self, p, frame, masks, others, key_out, val_out):
        K, _, H, W = masks.shape
        P = H * W
        planes = [(frame[0, c], 0) for c in range(3)] + [(masks, P), (others, P)]
        f16, _, _ = run_trunk_planes(p["menc"], planes, K, H, W, keep=False)
        return ops.conv(f16, p["kv_m"], out=key_out, out2=val_out)

    def segment(self, keys, values, q, logits=False, keys_split=None):
        """keys [K,n_mem,128], values [K,n_mem,512] (rows per memory position), q QueryFeatures ->
        object probabilities [K,H,W] (sigmoid applied, prop_net.py:181) or raw logits.  keys_split: ops.split_keys(keys) when
        the caller keeps one (InferenceCore's bank does); otherwise the memory read converts the keys on every call."""
        dec = self.plan()["dec"]
        K = keys.shape[0]
        _, h, w, _ = q.f16.shape
        cap = ops.max_act_batch(4 * h, 4 * w, 256)
        if K > cap:
            return torch.cat([self.segment(keys[i:i + cap], values[i:i + cap], q, logits,
                                           None if keys_split is None else keys_split[i:i + cap]) for i in range(0, K, cap)], 0)
        s8, s4 = self._skip(q)
        if ops.act_path():
            # the readout lands pre-split (x and relu(x)) in the compress block's input buffers; the v16 half of
            # cat([mem, v16.expand(K)]) (prop_net.py:178-179) enters as the cached partial sums q.c1v / q.dsv
            raw, rel = ops.memory_read_acts(keys, values, q.k16.reshape(h * w, CK), self.memory.top_k, h, w, keys_split=keys_split)
            cs = dec["compress_split"]
            x = run_resblock_acts((cs["c1m"], cs["c2"], cs["dsm"]), raw, rel, res1=q.c1v, res_skip=q.dsv)
        else:
            m4 = torch.empty((K, h, w, 2 * CV), dtype=torch.float32, device=keys.device)
            ops.memory_read(keys, values, q.k16.reshape(h * w, CK), self.memory.top_k, out=m4.view(K, h * w, 2 * CV)[:, :, :CV],
                            keys_split=keys_split)
            m4[..., CV:] = q.v16                                        # cat([mem, v16.expand(K)]) (prop_net.py:178-179)
            x = run_resblock(dec["compress"], m4)
        x = run_up_branch(dec["up_16_8"], s8, x, tag="up16")
        x = run_up_branch(dec["up_8_4"], s4, x, tag="up8")
        lo = ops.conv(x, dec["pred"], relu_in=True)                     # [K, H/4, W/4, 1]
        return ops.resize_bilinear(lo.view(K, lo.shape[1], lo.shape[2]), 4 * lo.shape[1], 4 * lo.shape[2],
                                   act=0 if logits else 1)

    def attention_lowres(self, mk, pos16, neg16, qk16):
        """mk [K,HW,128], pos16/neg16 [K,HW], qk16 [HW,128] -> [K,2,HW] aligned difference maps."""
        return ops.attention_align(mk, qk16, pos16, neg16)

    # ---- reference-compatible public API (logical NCHW) -----------------------------------
    # Every public method makes the device of its operands current for the duration of the call (the kernels launch
    # on the current HIP device's stream).
    def memorize(self, frame, masks):
        k, _, h, w = masks.shape
        with ops.on_device(masks):
            k16, v16 = self.memorize_into(frame.view(1, 3, h, w), masks)
            return k16.permute(0, 3, 1, 2).unsqueeze(2), v16.permute(0, 3, 1, 2).unsqueeze(2)

    def get_query_values(self, frame):
        with ops.on_device(frame):
            return self.encode_query(frame).as_reference_tuple()

    def segment_with_query(self, keys, values, f16, f8, f4, k16, v16):
        K, _, T, h, w = keys.shape
        with ops.on_device(keys):
            feats = [_nhwc(t) for t in (f16, f8, f4)]
            if ops.act_path():
                feats = [ops.to_act(t) for t in feats]
            q = QueryFeatures(*feats, _nhwc(k16), _nhwc(v16))
            kr = keys.permute(0, 2, 3, 4, 1).reshape(K, T * h * w, CK)
            vr = values.permute(0, 2, 3, 4, 1).reshape(K, T * h * w, CV)
            prob = self.segment(kr, vr, q)
            return prob.unsqueeze(1)

    def get_W(self, mk16, qk):
        """prop_net.py:183-185: the dense [HW x HW] attention matrix (softmax over the memory positions)."""
        return self.attn_memory(mk16, qk)

    def get_attention(self, mk16, pos_mask, neg_mask, qk16):
        b, _, h, w = pos_mask.shape
        nh, nw = h // 16, w // 16
        with ops.on_device(mk16):
            mk = mk16.permute(0, 2, 3, 4, 1).reshape(b, nh * nw, CK)
            qk = _nhwc(qk16).reshape(nh * nw, CK)
            pos = ops.area_pool16(pos_mask.reshape(b, h, w)).view(b, nh * nw)
            neg = ops.area_pool16(neg_mask.reshape(b, h, w)).view(b, nh * nw)
            low = self.attention_lowres(mk, pos, neg, qk)                   # [b, 2, nh*nw]
            return ops.resize_bilinear(low.view(b * 2, nh, nw), h, w).view(b, 2, h, w)

    def forward(self, *a, **k):
        raise NotImplementedError("use memorize / get_query_values / segment_with_query / get_attention")

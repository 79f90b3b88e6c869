"""CPU restatement of MiVOS's propagation + difference-aware-fusion hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file, and only as the checker; the
product (``mivos_amd``) never does and raises if its HIP library is missing.

Everything here is plain ``torch.nn.functional`` on CPU tensors over a flat
``state_dict`` (no nn.Module), in fp32 by default or fp64 as an arbitration "truth".
Each function cites the reference lines it restates (paths relative to
`/root/reference`).  PARITY PIN: the reference ships no tests or golden vectors for
this path, so the pin is the reference *itself* executed in this
container (oracle/ref_loader.py + oracle/make_golden.py); its outputs are committed
under tests/golden/ and tests/test_oracle_golden.py checks this file against them.
"""
import contextlib
import math

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------ primitive layers

TOPK_GAP = None  # set to a list to record top-k selection margins (see topk_softmax)
_CALIB = None   # when set (dict), BN measures + stores statistics (oracle/weights.calibrate)


@contextlib.contextmanager
def bn_calibration(store):
    global _CALIB
    _CALIB = store
    try:
        yield store
    finally:
        _CALIB = None


def _conv(sd, name, x, stride=1, pad=0, dilation=1):
    w = sd[name + ".weight"].to(x.dtype)
    b = sd.get(name + ".bias")
    y = F.conv2d(x, w, None if b is None else b.to(x.dtype), stride=stride, padding=pad, dilation=dilation)
    if _CALIB is not None and name in _CALIB.get("__lsuv__", {}):
        # weight-synthesis only (oracle/weights.calibrate): rescale this conv so that its
        # output has the requested std on the calibration clip; equals scaling weight+bias.
        g = float(_CALIB["__lsuv__"][name] / y.std())
        _CALIB["gain:" + name] = torch.tensor(g, dtype=torch.float32)
        y = y * g
    return y


def _bn(sd, name, x):
    """Eval-mode BatchNorm2d, eps 1e-5 (torch default; mod_resnet.py:85, torchvision)."""
    if _CALIB is not None and _CALIB.get("__bn__", True):
        m = x.mean(dim=(0, 2, 3))
        v = x.var(dim=(0, 2, 3), unbiased=True)
        _CALIB[name + ".running_mean"] = m.detach().float().clone()
        _CALIB[name + ".running_var"] = v.detach().float().clone()
        sd = dict(sd)
        sd[name + ".running_mean"], sd[name + ".running_var"] = m.float(), v.float()
    d = x.dtype
    return F.batch_norm(x, sd[name + ".running_mean"].to(d), sd[name + ".running_var"].to(d),
                        sd[name + ".weight"].to(d), sd[name + ".bias"].to(d), False, 0.0, 1e-5)


def _bottleneck(sd, p, x, stride):
    """mod_resnet.py:92-112 / torchvision Bottleneck: 1x1 -> 3x3(stride) -> 1x1, each +BN,
    ReLU after the first two, residual (optionally 1x1-stride conv + BN) then ReLU."""
    y = F.relu(_bn(sd, p + "bn1", _conv(sd, p + "conv1", x)))
    y = F.relu(_bn(sd, p + "bn2", _conv(sd, p + "conv2", y, stride=stride, pad=1)))
    y = _bn(sd, p + "bn3", _conv(sd, p + "conv3", y))
    if (p + "downsample.0.weight") in sd:
        x = _bn(sd, p + "downsample.1", _conv(sd, p + "downsample.0", x, stride=stride))
    return F.relu(y + x)


def _stage(sd, p, x, depth, stride):
    for b in range(depth):
        x = _bottleneck(sd, f"{p}.{b}.", x, stride if b == 0 else 1)
    return x


def _stem(sd, p, x):
    """conv 7x7/2 pad 3 -> BN -> ReLU -> maxpool 3/2/1 (modules.py:56-59, 81-84)."""
    x = F.relu(_bn(sd, p + "bn1", _conv(sd, p + "conv1", x, stride=2, pad=3)))
    return F.max_pool2d(x, 3, 2, 1)


def rgb_encoder(sd, frame):
    """RGBEncoder.forward, modules.py:80-89 -> (f16, f8, f4)."""
    p = "rgb_encoder."
    x = _stem(sd, p, frame)
    f4 = _stage(sd, p + "res2", x, 3, 1)
    f8 = _stage(sd, p + "layer2", f4, 4, 2)
    f16 = _stage(sd, p + "layer3", f8, 6, 2)
    return f16, f8, f4


def mask_rgb_encoder(sd, frame, mask, others):
    """MaskRGBEncoder.forward, modules.py:52-64: 5-channel input cat[f, m, o]."""
    p = "mask_rgb_encoder."
    x = _stem(sd, p, torch.cat([frame, mask, others], 1))
    x = _stage(sd, p + "layer1", x, 3, 1)
    x = _stage(sd, p + "layer2", x, 4, 2)
    return _stage(sd, p + "layer3", x, 6, 2)


def key_value(sd, p, x):
    """KeyValue.forward, modules.py:113-114."""
    return _conv(sd, p + "key_proj", x, pad=1), _conv(sd, p + "val_proj", x, pad=1)


def res_block(sd, p, x):
    """ResBlock.forward, modules.py:28-35 (pre-activation; 3x3 'downsample' on the skip)."""
    r = _conv(sd, p + "conv1", F.relu(x), pad=1)
    r = _conv(sd, p + "conv2", F.relu(r), pad=1)
    if (p + "downsample.weight") in sd:
        x = _conv(sd, p + "downsample", x, pad=1)
    return x + r


def upsample_block(sd, p, skip_f, up_f):
    """UpsampleBlock.forward, modules.py:100-104."""
    x = res_block(sd, p + "skip_conv2.", _conv(sd, p + "skip_conv1", skip_f, pad=1))
    x = x + F.interpolate(up_f, scale_factor=2, mode="bilinear", align_corners=False)
    return res_block(sd, p + "out_conv.", x)


def decoder(sd, f16, f8, f4):
    """Decoder.forward, prop_net.py:23-31 -> mask logits at full resolution."""
    x = res_block(sd, "decoder.compress.", f16)
    x = upsample_block(sd, "decoder.up_16_8.", f8, x)
    x = upsample_block(sd, "decoder.up_8_4.", f4, x)
    x = _conv(sd, "decoder.pred", F.relu(x), pad=1)
    return F.interpolate(x, scale_factor=4, mode="bilinear", align_corners=False)

# ------------------------------------------------------------------ memory read


def affinity(mk, qk):
    """prop_net.py:85-88: A[b, m, q] = sum_c mk[b,c,m] * (qk[c,q] / sqrt(CK)), m over T*H*W."""
    B, CK = mk.shape[:2]
    mi = mk.reshape(B, CK, -1).transpose(1, 2)
    qi = qk.reshape(1, CK, -1).expand(B, -1, -1) / math.sqrt(CK)
    return torch.bmm(mi, qi)


def topk_softmax(aff, top_k):
    """softmax_w_g_top(gauss=None), prop_net.py:54-63.  NOTE `values[:, 0]` is the
    per-column maximum only for B == 1, which is how the reference calls it
    (segment_with_query batched=1, prop_net.py:173-176)."""
    assert aff.shape[0] == 1
    values, indices = torch.topk(aff, k=top_k, dim=1)
    if TOPK_GAP is not None and aff.shape[1] > top_k:
        # conditioning probe (make_golden.py): smallest margin between the k-th and (k+1)-th score;
        # a margin at rounding level means top-k membership is decided by fp32 noise
        v2 = torch.topk(aff, k=top_k + 1, dim=1)[0]
        TOPK_GAP.append(float((v2[:, top_k - 1] - v2[:, top_k]).min()))
    e = torch.exp(values - values[:, 0])
    e = e / e.sum(dim=1, keepdim=True)
    return torch.zeros_like(aff).scatter_(1, indices, e), values, indices


def memory_read(mk, mv, qk, top_k):
    """EvalMemoryReader.forward (km=None), prop_net.py:81-108.  mk [1,CK,T,H,W],
    mv [1,CV,T,H,W], qk [1,CK,H,W] -> [1,CV,H,W]."""
    B, CV = mv.shape[:2]
    H, W = qk.shape[-2:]
    a = affinity(mk, qk)
    if top_k is not None:
        a, _, _ = topk_softmax(a, top_k)
    else:
        a = F.softmax(a, dim=1)
    return torch.bmm(mv.reshape(B, CV, -1), a).view(B, CV, H, W)


def get_attention(mk16, pos_mask, neg_mask, qk16):
    """PropagationNetwork.get_attention + AttentionMemory.forward, prop_net.py:115-129,187-200."""
    b, _, h, w = pos_mask.shape
    nh, nw = h // 16, w // 16
    W = F.softmax(affinity(mk16, qk16), dim=1)
    pos = F.interpolate(pos_mask, size=(nh, nw), mode="area").view(b, 1, nh * nw) @ W
    neg = F.interpolate(neg_mask, size=(nh, nw), mode="area").view(b, 1, nh * nw) @ W
    a = torch.cat([pos, neg], 1).reshape(b, 2, nh, nw)
    return F.interpolate(a, mode="bilinear", size=(h, w), align_corners=False)

def attention_weights(mk, qk):
    """AttentionMemory.forward of model/attn_network.py:17-28: mk, qk [B,CK,H,W] -> W [B,HW,HW], softmax over the
    memory positions (dim 1); every sample has its own query map."""
    B, CK = mk.shape[:2]
    mi = mk.reshape(B, CK, -1).transpose(1, 2)
    qi = qk.reshape(B, CK, -1) / math.sqrt(CK)
    return F.softmax(torch.bmm(mi, qi), dim=1)


def attention_read_network(sd, image, mask11, mask21, mask12, mask22, query_image):
    """AttentionReadNetwork.forward, model/attn_network.py:46-80 (same encoders / KeyValue weights as the propagation
    network): aligned positive / negative difference maps of two objects."""
    b, _, h, w = mask11.shape
    nh, nw = h // 16, w // 16
    pos1, neg1 = (mask21 - mask11).clamp(0, 1), (mask11 - mask21).clamp(0, 1)
    pos2, neg2 = (mask22 - mask12).clamp(0, 1), (mask12 - mask22).clamp(0, 1)
    k1, _ = key_value(sd, "kv_m_f16.", mask_rgb_encoder(sd, image, mask21, mask22))
    k2, _ = key_value(sd, "kv_m_f16.", mask_rgb_encoder(sd, image, mask22, mask21))
    qk16, _ = key_value(sd, "kv_q_f16.", rgb_encoder(sd, query_image)[0])
    outs = []
    for keys, pos, neg in ((k1, pos1, neg1), (k2, pos2, neg2)):
        W = attention_weights(keys, qk16)
        pm = F.interpolate(pos, size=(nh, nw), mode="area").view(b, 1, nh * nw) @ W
        nm = F.interpolate(neg, size=(nh, nw), mode="area").view(b, 1, nh * nw) @ W
        a = torch.cat([pm, nm], 1).reshape(b, 2, nh, nw)
        outs.append(F.interpolate(a, mode="bilinear", size=(h, w), align_corners=False))
    return outs[0], outs[1]


def aggregate_wbg_channel(prob, keep_bg=False, hard=False):
    """model/aggregate.py:39-53: prob [B,K,H,W] -> (logits [B,K+1,H,W], softmax over dim 1)."""
    p = torch.cat([torch.prod(1 - prob, dim=1, keepdim=True), prob], 1).clamp(1e-7, 1 - 1e-7)
    logits = torch.log(p / (1 - p))
    if hard:
        logits = logits * 1000
    s = F.softmax(logits, dim=1)
    return logits, (s if keep_bg else s[:, 1:])

# ------------------------------------------------------------------ network methods


def memorize(sd, frame, masks):
    """PropagationNetwork.memorize, prop_net.py:144-162 -> k [K,128,1,h,w], v [K,512,1,h,w]."""
    k = masks.shape[0]
    frames = frame.reshape(1, 3, *masks.shape[-2:]).repeat(k, 1, 1, 1)
    if k != 1:
        others = torch.cat([masks[[j for j in range(k) if j != i]].sum(0, keepdim=True) for i in range(k)], 0)
    else:
        others = torch.zeros_like(masks)
    k16, v16 = key_value(sd, "kv_m_f16.", mask_rgb_encoder(sd, frames, masks, others))
    return k16.unsqueeze(2), v16.unsqueeze(2)


def get_query_values(sd, frame):
    """prop_net.py:164-168."""
    f16, f8, f4 = rgb_encoder(sd, frame)
    k16, v16 = key_value(sd, "kv_q_f16.", f16)
    return f16, f8, f4, k16, v16


def segment_logits(sd, keys, values, f16, f8, f4, k16, v16, top_k):
    """segment_with_query before the sigmoid, prop_net.py:170-181 (object by object)."""
    k = keys.shape[0]
    m4 = torch.cat([memory_read(keys[i:i + 1], values[i:i + 1], k16, top_k) for i in range(k)], 0)
    m4 = torch.cat([m4, v16.expand(k, -1, -1, -1)], 1)
    return decoder(sd, m4, f8, f4)


def segment_with_query(sd, keys, values, f16, f8, f4, k16, v16, top_k):
    return torch.sigmoid(segment_logits(sd, keys, values, f16, f8, f4, k16, v16, top_k))


def fusion_net(fsd, im, seg1, seg2, attn, time):
    """FusionNet.forward, fusion_net.py:32-50 (returns the logit)."""
    h, w = im.shape[-2:]
    t = time.to(im.dtype)[:, :, None, None].expand(-1, -1, h, w)
    x = torch.cat([im, seg1, seg2, attn, t], 1)
    x = F.relu(_conv(fsd, "conv1.0", x, pad=1))
    for blk in ("conv2", "conv3"):
        r = _conv(fsd, blk + ".2", F.relu(_conv(fsd, blk + ".0", x, pad=1)), pad=1)
        x = F.relu(x + r)
    return _conv(fsd, "final_conv", x, pad=1)


def aggregate_wbg(prob, keep_bg=False, hard=False):
    """model/aggregate.py:22-37."""
    p = torch.cat([torch.prod(1 - prob, dim=0, keepdim=True), prob], 0).clamp(1e-7, 1 - 1e-7)
    logits = torch.log(p / (1 - p))
    if hard:
        logits = logits * 1000
    s = F.softmax(logits, dim=0)
    return s if keep_bg else s[1:]


def aggregate_sbg(prob, keep_bg=False, hard=False):
    """model/aggregate.py:4-20 (background fixed at 0.5)."""
    k, _, h, w = prob.shape
    p = torch.cat([torch.full((1, 1, h, w), 0.5, dtype=prob.dtype), prob], 0).clamp(1e-7, 1 - 1e-7)
    logits = torch.log(p / (1 - p))
    if hard:
        logits = logits * 1000
    s = F.softmax(logits, dim=0)
    return s if keep_bg else s[1:]


def pad_divide_by(x, d):
    """util/tensor_util.py:62-80: symmetric zero pad to a multiple of d, low side gets
    floor(delta/2)."""
    h, w = x.shape[-2:]
    nh, nw = (h + d - 1) // d * d, (w + d - 1) // d * d
    lh, lw = (nh - h) // 2, (nw - w) // 2
    pad = (lw, nw - w - lw, lh, nh - h - lh)
    return F.pad(x, pad), pad

# ------------------------------------------------------------------ InferenceCore


class OracleCore:
    """Restatement of InferenceCore (inference_core.py:17-293) for mem_profile=0 on CPU.

    ``trace`` records the schedule in SURVEY.md §3's notation: 'M' memorize, 'Q' query
    encode (cache miss), 'S<n>' segment with n memory frames, 'F(nc,nr)' one FusionNet
    call.  ``logits`` keeps the decoder logits of the last pass per frame for
    tolerance checks."""

    def __init__(self, sd, fsd, images, num_objects, mem_freq=5, top_k=50, dtype=torch.float32, record_margins=False):
        self.sd, self.fsd, self.top_k, self.dtype = sd, fsd, top_k, dtype
        # record_margins: per propagated frame the smallest gap between the k-th and (k+1)-th affinity of any query / object
        # (test diagnostics: a gap below the rounding noise of the keys means top-k membership is decided by that noise)
        self.record_margins, self.topk_margin = record_margins, {}
        self.mem_freq, self.k = mem_freq, num_objects
        self.t = images.shape[1]
        self.h, self.w = images.shape[-2:]
        self.images, self.pad = pad_divide_by(images.to(dtype), 16)          # :71
        self.nh, self.nw = self.images.shape[-2:]
        self.prob = torch.zeros((self.k + 1, self.t, 1, self.nh, self.nw), dtype=dtype)
        self.prob[0] = 1e-7                                                    # :82
        self.masks = torch.zeros((self.t, 1, self.nh, self.nw), dtype=torch.uint8)
        self.np_masks = np.zeros((self.t, self.h, self.w), dtype=np.uint8)
        self.query_buf, self.interacted = {}, set()
        self.certain_k = self.certain_v = None
        self.trace, self.logits, self.propagated = [], {}, 0
        # step_hook(record): called once per propagated frame with everything that went into and came out of the step (bank
        # views, query features, decoder logits, aggregated output, the frame's memorised key / value, the fusion branch's
        # inputs and logits) - the tests run the engine on the SAME inputs ("teacher forced") and compare each stage
        self.step_hook = None

    def _query(self, ti):                                                      # :110-120
        if ti not in self.query_buf:
            self.trace.append("Q")
            self.query_buf[ti] = get_query_values(self.sd, self.images[:, ti])
        return self.query_buf[ti]

    def _memorize(self, ti, masks):
        self.trace.append("M")
        return memorize(self.sd, self.images[:, ti], masks)

    def do_pass(self, key_k, idx, forward=True, step_cb=None):                 # :122-200
        nc = self.certain_k.shape[2]
        m_front = nc
        if forward:
            closest = min([t for t in self.interacted if t > idx] + [self.t])
            n = closest - idx - 1
            rng, end = range(idx + 1, closest), closest - 1
        else:
            closest = max([t for t in self.interacted if t < idx] + [-1])
            n = idx - closest - 1
            rng, end = range(idx - 1, closest, -1), closest + 1
        total_m = n // self.mem_freq + 1 + nc
        K, CK, _, H, W = key_k.shape
        keys = torch.empty((K, CK, total_m, H, W), dtype=self.dtype)
        values = torch.empty((K, 512, total_m, H, W), dtype=self.dtype)
        keys[:, :, :nc], values[:, :, :nc] = self.certain_k, self.certain_v
        prev_in_mem, last_ti = True, idx
        fuse = closest != self.t and closest != -1
        for ti in rng:
            m = m_front if prev_in_mem else m_front + 1
            q = self._query(ti)
            self.trace.append(f"S{m}")
            global TOPK_GAP
            if self.record_margins:
                TOPK_GAP = []
            logit = segment_logits(self.sd, keys[:, :, :m], values[:, :, :m], *q, self.top_k)
            if self.record_margins:
                self.topk_margin[ti] = min(TOPK_GAP) if TOPK_GAP else float("inf")
                TOPK_GAP = None
            self.logits[ti] = logit
            out = aggregate_wbg(torch.sigmoid(logit), keep_bg=True)
            rec = None
            if self.step_hook is not None:
                rec = dict(ti=ti, idx=idx, closest=closest, n_mem=m, keys=keys[:, :, :m].clone(), values=values[:, :, :m].clone(), query=q,
                           logit=logit, out=out, memorized=None, fuse=None)
            if ti != end:
                mem_kv = self._memorize(ti, out[1:])
                keys[:, :, m_front:m_front + 1], values[:, :, m_front:m_front + 1] = mem_kv
                if rec is not None:
                    rec["memorized"] = mem_kv
                if abs(ti - last_ti) >= self.mem_freq:
                    m_front, last_ti, prev_in_mem = m_front + 1, ti, True
                else:
                    prev_in_mem = False
            if fuse:
                prev = self.prob[:, ti].clone()
                self.prob[:, ti] = self.fuse_one_frame(closest, idx, ti, prev, out, key_k, q[3])
                if rec is not None:
                    rec["fuse"] = dict(prev=prev, curr=out, key_k=key_k, qk16=q[3], logits=self.last_fuse_logits, attn=self.last_fuse_attn,
                                       fused=self.prob[:, ti].clone())
            else:
                self.prob[:, ti] = out
            if rec is not None:
                self.step_hook(rec)
            self.propagated += 1
            if step_cb is not None:                                                # :195-196
                step_cb()
        return closest

    def fuse_one_frame(self, tc, tr, ti, prev, curr, mk16, qk16):              # :202-217
        assert tc < ti < tr or tr < ti < tc
        nc, nr = abs(tc - ti) / abs(tc - tr), abs(tr - ti) / abs(tc - tr)
        dist = torch.tensor([[nc, nr]], dtype=torch.float32).to(self.dtype)
        prob = torch.zeros((self.k, 1, self.nh, self.nw), dtype=self.dtype)
        logits, attns = [], []
        for k in range(1, self.k + 1):
            attn = get_attention(mk16[k - 1:k], self.pos_diff[k:k + 1], self.neg_diff[k:k + 1], qk16)
            self.trace.append(f"F({nc:.2f},{nr:.2f})")
            z = fusion_net(self.fsd, self.images[:, ti], prev[k:k + 1], curr[k:k + 1], attn, dist)
            logits.append(z)
            attns.append(attn)
            prob[k - 1] = torch.sigmoid(z)
        self.last_fuse_logits, self.last_fuse_attn = torch.cat(logits, 0), torch.cat(attns, 0)      # [K,1,nh,nw], [K,2,nh,nw]
        return aggregate_wbg(prob, keep_bg=True)

    def interact(self, mask, idx, total_cb=None, step_cb=None):                # :219-271
        self.interacted.add(idx)
        mask, _ = pad_divide_by(mask.to(self.dtype), 16)
        diff = mask - self.prob[:, idx]
        self.pos_diff, self.neg_diff = diff.clamp(0, 1), (-diff).clamp(0, 1)
        self.prob[:, idx] = mask
        key_k, key_v = self._memorize(idx, mask[1:])
        if self.certain_k is None:
            self.certain_k, self.certain_v = key_k, key_v
        else:
            self.certain_k = torch.cat([self.certain_k, key_k], 2)
            self.certain_v = torch.cat([self.certain_v, key_v], 2)
        if total_cb is not None:                                                   # :247-253
            front = min([ti for ti in self.interacted if ti > idx] + [self.t])
            back = max([ti for ti in self.interacted if ti < idx] + [-1])
            if front - back - 2 > 0:
                total_cb(front - back - 2)
        self.do_pass(key_k, idx, True, step_cb)
        self.do_pass(key_k, idx, False, step_cb)
        for ti in range(self.t):
            self.masks[ti] = torch.argmax(self.prob[:, ti], dim=0)
        lw, uw, lh, uh = self.pad
        out = self.masks[:, 0, lh:self.nh - uh, lw:self.nw - uw]
        self.np_masks = out.numpy().astype(np.uint8)
        return self.np_masks

    def update_mask_only(self, prob_mask, idx):                                # :273-293
        """Interaction without propagation: argmax over the K+1 channels of the (padded) prob_mask -> masks[idx], cropped ->
        np_masks[idx]; every other frame keeps its result."""
        mask = torch.argmax(prob_mask, 0)
        self.masks[idx] = mask
        lw, uw, lh, uh = self.pad
        self.np_masks[idx] = mask[0, lh:self.nh - uh, lw:self.nw - uw].numpy().astype(np.uint8)
        return self.np_masks


class OracleGenerator:
    """Restatement of FusionGenerator (generation/fusion_generator.py:12-101): propagation from one annotated frame to both range
    limits, no fusion, no query cache; the bank grows by torch.cat with a temporary entry for the previous frame."""

    def __init__(self, sd, images, mem_freq, top_k=50, dtype=torch.float32, record_margins=False):
        self.sd, self.mem_freq, self.top_k, self.dtype = sd, mem_freq, top_k, dtype
        self.record_margins, self.topk_margin = record_margins, {}
        self.t = images.shape[1]
        self.h, self.w = images.shape[-2:]
        self.images, self.pad = pad_divide_by(images.to(dtype), 16)                       # :22
        self.nh, self.nw = self.images.shape[-2:]

    def reset(self, k):                                                                    # :32-34
        self.k = k
        self.prob = torch.zeros((k + 1, self.t, 1, self.nh, self.nw), dtype=self.dtype)

    def do_pass(self, key_k, key_v, idx, left_limit, right_limit, forward=True):            # :43-78
        keys, values = key_k, key_v
        prev_k = prev_v = None
        last_ti = idx
        if forward:
            rng, end = range(idx + 1, right_limit + 1), right_limit
        else:
            rng, end = range(idx - 1, left_limit - 1, -1), left_limit
        for ti in rng:
            this_k = keys if prev_k is None else torch.cat([keys, prev_k], 2)
            this_v = values if prev_v is None else torch.cat([values, prev_v], 2)
            q = get_query_values(self.sd, self.images[:, ti])
            global TOPK_GAP
            if self.record_margins:
                TOPK_GAP = []
            out = aggregate_wbg(segment_with_query(self.sd, this_k, this_v, *q, top_k=self.top_k), keep_bg=True)
            if self.record_margins:
                self.topk_margin[ti] = min(TOPK_GAP) if TOPK_GAP else float("inf")
                TOPK_GAP = None
            self.prob[:, ti] = out
            if ti != end:
                prev_k, prev_v = memorize(self.sd, self.images[:, ti], out[1:])
                if abs(ti - last_ti) >= self.mem_freq:
                    last_ti = ti
                    keys, values = torch.cat([keys, prev_k], 2), torch.cat([values, prev_v], 2)
                    prev_k = prev_v = None

    def interact_mask(self, mask, idx, left_limit, right_limit):                           # :80-101
        mask, _ = pad_divide_by(mask.to(self.dtype), 16)
        mask = aggregate_wbg(mask, keep_bg=True)
        self.prob[:, idx] = mask
        key_k, key_v = memorize(self.sd, self.images[:, idx], mask[1:])
        self.do_pass(key_k, key_v, idx, left_limit, right_limit, True)
        self.do_pass(key_k, key_v, idx, left_limit, right_limit, False)
        lw, uw, lh, uh = self.pad
        return self.prob[:, :, 0, lh:self.nh - uh, lw:self.nw - uw]


# ------------------------------------------------------------------ synthetic clips
from mivos_amd.util.synthetic import synthetic_clip  # noqa: E402,F401  (shared input generator)

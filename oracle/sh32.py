"""CPU restatement of the SH32 operand formats of the LDS-DMA convolution path (test infrastructure only).

Nothing in the reference corresponds to these layouts - they are an internal storage format of mivos_amd
(mivos_amd/csrc/conv_f16x3_dma.hip) for the tensors that travel between the convolutions of
model/propagation/modules.py / mod_resnet.py - so the "oracle" here is the documented definition itself, written
with plain torch CPU ops; tests/test_gpu_ops.py compares the HIP pack kernels with it bit for bit.
"""
import torch


def split_hi_lo(x):
    """fp32 -> (hi, lo) fp16 with x ~= hi + lo: hi = fp16(x) (round to nearest even), lo = fp16(x - hi)."""
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return hi, lo


def pack_activation(x_nhwc, relu=False, border=1):
    """fp32 [N,H,W,C] (C % 32 == 0) -> float32-typed buffer [N, H+2b, W+2b, C] holding, per pixel and group of 32
    channels, 32 fp16 hi parts followed by 32 fp16 lo parts; border pixels are zero."""
    n, h, w, c = x_nhwc.shape
    assert c % 32 == 0
    x = torch.relu(x_nhwc) if relu else x_nhwc
    hi, lo = split_hi_lo(x.float())
    lines = torch.stack([hi.view(n, h, w, c // 32, 32), lo.view(n, h, w, c // 32, 32)], dim=4)     # [N,H,W,G,2,32] halves
    buf = torch.zeros(n, h + 2 * border, w + 2 * border, c // 32, 2, 32, dtype=torch.float16)
    buf[:, border:border + h, border:border + w] = lines
    return buf.view(n, h + 2 * border, w + 2 * border, 2 * c).view(torch.float32)                  # 2C halves = C floats


def unpack_activation(buf, border=1):
    """Inverse of pack_activation: x = hi + lo (fp32)."""
    n, hp, wp, c = buf.shape
    halves = buf.contiguous().view(torch.float16).view(n, hp, wp, c // 32, 2, 32)
    x = halves[..., 0, :].float() + halves[..., 1, :].float()
    return x.reshape(n, hp, wp, c)[:, border:hp - border, border:wp - border]


def pack_weights_dma(w_ohwi, mult):
    """OHWI fp32 [Cout,KH,KW,Cin] -> uint8 buffer: 128 zero bytes, then for K step s = (c // 32) * taps + tap and output
    channel n one 128-byte line of eight 16-byte chunks; logical chunk lc = part * 4 + j (part 0: hi, 1: lo; j: channels
    8j..8j+7 of the slab) is stored at chunk position lc ^ ((n >> 1) & 7)."""
    cout, kh, kw, cin = w_ohwi.shape
    taps, slabs = kh * kw, cin // 32
    hi, lo = split_hi_lo(w_ohwi.float() * mult)
    # [Cout, taps, slabs, 32] -> [slabs, taps, Cout, part, 4 chunks, 8]
    def arrange(t):
        return t.view(cout, taps, slabs, 4, 8).permute(2, 1, 0, 3, 4)
    logical = torch.stack([arrange(hi), arrange(lo)], dim=3).reshape(slabs * taps, cout, 8, 8)    # [s][n][lc][8 halves]
    phys = torch.empty_like(logical)
    n = torch.arange(cout)
    for q in range(8):
        lc = q ^ ((n >> 1) & 7)                                                                    # which logical chunk sits at q
        phys[:, n, q] = logical[:, n, lc]
    body = phys.contiguous().view(torch.uint8).reshape(-1)
    return torch.cat([torch.zeros(128, dtype=torch.uint8), body])


def split_key_rows(keys):
    """fp32 key rows [..., 128] -> the rows mivos_memory_read_select_f16x3 streams (mivos_memory_split_keys), as a
    float32-typed tensor of the same shape: block b = 0..3 (64 halves) holds, for ks = 0..3, hi[8] | lo[8] of channels
    32 ks + 8 b + e."""
    assert keys.shape[-1] == 128
    hi, lo = split_hi_lo(keys.float())
    lead = keys.shape[:-1]
    def arrange(t):                                                  # [..., ks, b, e] -> [..., b, ks, e]
        return t.reshape(*lead, 4, 4, 8).transpose(-3, -2)
    rows = torch.stack([arrange(hi), arrange(lo)], dim=-2)           # [..., b, ks, part, e]
    return rows.reshape(*lead, 256).contiguous().view(torch.float32)


def affinity_f16x3(keys, qk):
    """What the f16x3 affinity computes, in fp64: keys [n_mem,128], qk [n_q,128] -> [n_mem, n_q] =
    sum_c (kh qh + kh ql + kl qh), k = kh + kl, q / sqrt(128) = qh + ql (the kl ql term is dropped)."""
    kh, kl = (t.double() for t in split_hi_lo(keys.float()))
    qh, ql = (t.double() for t in split_hi_lo(qk.float() / torch.sqrt(torch.tensor(128.0))))
    return kh @ qh.t() + kh @ ql.t() + kl @ qh.t()

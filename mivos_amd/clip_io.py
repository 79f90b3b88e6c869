"""Clip ingest and mask egress on either side of the propagation path.

Ingest (reference: `dataset/davis_test_dataset.py:66-110`, `dataset/yv_test_dataset.py:54-119`): decoded uint8 frames are
uploaded once (3 bytes per pixel instead of the 12 of a host-side float tensor) and normalised / resized / padded by HIP
kernels straight into the `[1,T,3,nh,nw]` layout ``InferenceCore`` keeps; the ground-truth label map of the annotated frame
becomes the one-hot mask ``interact`` expects.  Egress (reference: `eval_interactive_davis.py:86-94`): ``InferenceCore``
already returns the whole clip's palette-index masks with one device-to-host copy; ``write_palette_png`` stores them as
indexed PNGs (zlib + struct, no imaging library needed) like the reference's ``Image.putpalette(...).save(...)``.
"""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

IM_MEAN = (0.485, 0.456, 0.406)        # dataset/range_transform.py:5-8
IM_STD = (0.229, 0.224, 0.225)


def _pad16(h, w):
    """pad_divide_by(.., 16) geometry (util/tensor_util.py:62-80): (nh, nw, (left, right, top, bottom))."""
    dh, dw = (-h) % 16, (-w) % 16
    return h + dh, w + dw, (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2)


def yv_480p_size(h, w):
    """yv_test_dataset.py:102-109: short side -> 480."""
    return (h * 480 // w, 480) if h > w else (480, w * 480 // h)


def ingest_frames(frames_u8, device="cuda:0", resize_to=None, padded=True):
    """frames_u8: uint8 [T,H,W,3] (numpy or torch, as decoded) -> normalised float32 images on `device`.

    padded=True returns ([1,T,3,nh,nw] zero-padded to multiples of 16, pad) - exactly what InferenceCore builds from the
    reference's CPU tensor (same values bit for bit); padded=False returns the unpadded [1,T,3,H',W'].  resize_to=(H',W')
    applies the YouTube-VOS loader's bicubic resize first."""
    fr = torch.as_tensor(np.ascontiguousarray(frames_u8)) if not isinstance(frames_u8, torch.Tensor) else frames_u8.contiguous()
    assert fr.dtype == torch.uint8 and fr.dim() == 4 and fr.shape[3] == 3
    T, H, W, _ = fr.shape
    dev = torch.device(device)
    with ops.on_device(dev):
        fr = fr.to(dev, non_blocking=True)
        lib = _lib.load()
        mean, std = (C.c_float * 3)(*IM_MEAN), (C.c_float * 3)(*IM_STD)
        oh, ow = (H, W) if resize_to is None else resize_to
        nh, nw, pad = _pad16(oh, ow) if padded else (oh, ow, (0, 0, 0, 0))
        out = torch.zeros((1, T, 3, nh, nw), dtype=torch.float32, device=dev)
        if resize_to is None:
            check(lib.mivos_ingest_u8(fr.data_ptr(), out.data_ptr(), T, H, W, 3 * nh * nw, nh * nw, nw, pad[2], pad[0], mean, std, ops._stream()))
        else:
            tmp = torch.empty((T, 3, H, W), dtype=torch.float32, device=dev)
            check(lib.mivos_ingest_u8(fr.data_ptr(), tmp.data_ptr(), T, H, W, 3 * H * W, H * W, W, 0, 0, mean, std, ops._stream()))
            check(lib.mivos_resize_bicubic(tmp.data_ptr(), out.data_ptr(), T * 3, H, W, oh, ow, nh * nw, nw, pad[2], pad[0], ops._stream()))
    return (out, pad) if padded else out


def onehot_mask(label_map_u8, labels, device="cuda:0", resize_to=None):
    """Palette-index label map uint8 [H,W] -> float one-hot [K+1,1,H',W'] (background first: everything not in `labels`),
    the `mask` argument of InferenceCore.interact; resize_to applies the loader's nearest-neighbour resize."""
    lab = torch.as_tensor(np.ascontiguousarray(label_map_u8))
    assert lab.dtype == torch.uint8 and lab.dim() == 2
    H, W = lab.shape
    oh, ow = (H, W) if resize_to is None else resize_to
    dev = torch.device(device)
    with ops.on_device(dev):
        lab = lab.to(dev)
        lv = torch.as_tensor(np.asarray(labels, dtype=np.uint8)).to(dev)
        k = lv.numel()
        out = torch.empty((k + 1, 1, oh, ow), dtype=torch.float32, device=dev)
        check(_lib.load().mivos_onehot_nearest(lab.data_ptr(), lv.data_ptr(), k, out.data_ptr(), H, W, oh, ow, oh * ow, ow, 0, 0, ops._stream()))
    return out


def onehot_masks(label_maps_u8, labels, device="cuda:0", resize_to=None):
    """A clip's label maps uint8 [T,H,W] -> float one-hot [K+1,T,1,H',W'] (background first): ONE upload of the stack, one launch per
    frame writing straight into the result (the loaders' `gt` is `[1:]` of it).  resize_to: the loader's nearest-neighbour resize."""
    lab = torch.as_tensor(np.ascontiguousarray(label_maps_u8))
    assert lab.dtype == torch.uint8 and lab.dim() == 3
    T, H, W = lab.shape
    oh, ow = (H, W) if resize_to is None else resize_to
    dev = torch.device(device)
    with ops.on_device(dev):
        lab = lab.to(dev)
        lv = torch.as_tensor(np.asarray(labels, dtype=np.uint8)).to(dev)
        k = lv.numel()
        out = torch.empty((k + 1, T, 1, oh, ow), dtype=torch.float32, device=dev)
        lib = _lib.load()
        for t in range(T):
            check(lib.mivos_onehot_nearest(lab.data_ptr() + t * H * W, lv.data_ptr(), k, out.data_ptr() + 4 * t * oh * ow, H, W, oh, ow, T * oh * ow, ow, 0, 0,
                                           ops._stream()))
    return out


def _png_chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_palette_png(mask_u8, palette):
    """uint8 [H,W] of palette indices + palette (flat list of up to 256 RGB triples, as PIL's getpalette()) -> PNG bytes
    (8-bit indexed colour)."""
    m = np.ascontiguousarray(mask_u8, dtype=np.uint8)
    h, w = m.shape
    pal = bytes(bytearray(int(v) & 255 for v in palette))[:768]
    pal += b"\x00" * (-len(pal) % 3)
    raw = b"".join(b"\x00" + m[y].tobytes() for y in range(h))
    return (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 3, 0, 0, 0)) + _png_chunk(b"PLTE", pal) +
            _png_chunk(b"IDAT", zlib.compress(raw, 6)) + _png_chunk(b"IEND", b""))


def write_palette_png(np_masks, palette, out_dir, pattern="{:05d}.png"):
    """The reference's mask egress (eval_interactive_davis.py:86-94): one indexed PNG per frame of the uint8 [T,H,W] result."""
    os.makedirs(out_dir, exist_ok=True)
    for i, m in enumerate(np_masks):
        with open(os.path.join(out_dir, pattern.format(i)), "wb") as f:
            f.write(encode_palette_png(m, palette))

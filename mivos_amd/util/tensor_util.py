"""Padding / IoU helpers with the reference's `util/tensor_util.py` names and semantics
(pure tensor plumbing: no arithmetic kernels needed)."""
import numpy as np
import torch
import torch.nn.functional as F

_EPS = 1e-6


def _iu(seg, gt):
    return (seg & gt).sum(), (seg | gt).sum()


def compute_tensor_iu(seg, gt):
    i, u = _iu(seg, gt)
    return i.float(), u.float()


def compute_np_iu(seg, gt):
    i, u = _iu(seg, gt)
    return np.float32(i), np.float32(u)


def compute_tensor_iou(seg, gt):
    i, u = compute_tensor_iu(seg, gt)
    return (i + _EPS) / (u + _EPS)


def compute_np_iou(seg, gt):
    i, u = compute_np_iu(seg, gt)
    return (i + _EPS) / (u + _EPS)


def compute_multi_class_iou(seg, gt):
    """seg [K+1,H,W] scores (background first), gt [K,1,H,W]."""
    pred = torch.argmax(seg, dim=0)
    n = gt.shape[0]
    total = sum(compute_tensor_iou(pred == (k + 1), gt[k, 0] > 0.5) for k in range(n))
    return (total + _EPS) / (n + _EPS)


def compute_multi_class_iou_idx(seg, gt):
    """seg [H,W] label map, gt [K,H,W]."""
    n = gt.shape[0]
    total = sum(compute_np_iou(seg == (k + 1), gt[k] > 0.5) for k in range(n))
    return (total + _EPS) / (n + _EPS)


def compute_multi_class_iou_both_idx(seg, gt):
    n = gt.max()
    total = sum(compute_np_iou(seg == k, gt == k) for k in range(1, n + 1))
    return (total + _EPS) / (n + _EPS)


def pad_divide_by(in_img, d, in_size=None):
    """Zero-pad the last two dims symmetrically up to multiples of d; the low side gets
    floor(delta / 2).  Returns (padded, (left, right, top, bottom))."""
    h, w = in_img.shape[-2:] if in_size is None else in_size
    dh, dw = (-h) % d, (-w) % d
    pad = (dw // 2, dw - dw // 2, dh // 2, dh - dh // 2)
    return F.pad(in_img, pad), pad


def _crop(img, pad, hdim, wdim):
    l, r, t, b = pad
    if t + b > 0:
        img = img.narrow(hdim, t, img.shape[hdim] - t - b)
    if l + r > 0:
        img = img.narrow(wdim, l, img.shape[wdim] - l - r)
    return img


def unpad(img, pad):
    return _crop(img, pad, 2, 3)


def unpad_3dim(img, pad):
    return _crop(img, pad, 1, 2)

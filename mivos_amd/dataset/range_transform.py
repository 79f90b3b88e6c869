"""`dataset/range_transform.py:1-12` of the reference without torchvision: the ImageNet normalisation of the test-time loaders
as plain callables on float CHW tensors (same arithmetic: sub_ then div_ by per-channel fp32 constants)."""
import torch

im_mean = (124, 116, 104)
IM_MEAN = (0.485, 0.456, 0.406)
IM_STD = (0.229, 0.224, 0.225)


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        t = t.clone()
        mean = torch.as_tensor(self.mean, dtype=t.dtype, device=t.device)[:, None, None]
        std = torch.as_tensor(self.std, dtype=t.dtype, device=t.device)[:, None, None]
        return t.sub_(mean).div_(std)


im_normalization = _Normalize(IM_MEAN, IM_STD)
inv_im_trans = _Normalize([-0.485 / 0.229, -0.456 / 0.224, -0.406 / 0.225], [1 / 0.229, 1 / 0.224, 1 / 0.225])

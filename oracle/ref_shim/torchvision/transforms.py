"""Test-infrastructure stub of the few `torchvision.transforms` the reference's test-time loaders use
(`dataset/range_transform.py:5-12`, `dataset/davis_test_dataset.py:49-62`, `dataset/yv_test_dataset.py:37-40`).  torchvision
(pinned 0.8.2 by the reference's README) is not installed in this image; these restate its published behaviour:

  ToTensor   PIL RGB / uint8 HWC -> float32 CHW in [0, 1]:  tensor(bytes).permute(2, 0, 1).float().div(255)
  Normalize  (x - mean[:, None, None]) / std[:, None, None] with mean / std as tensors of x's dtype (sub_ then div_)
  Compose    left-to-right application
  Resize     only constructed for resolutions other than 480p (never on the paths exercised here): raises when called

NOT shipped and NOT used by the product path."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    def __call__(self, pic):
        a = np.array(pic, np.uint8, copy=True)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(a).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = mean, std

    def __call__(self, t):
        t = t.clone()
        mean = torch.as_tensor(self.mean, dtype=t.dtype)[:, None, None]
        std = torch.as_tensor(self.std, dtype=t.dtype)[:, None, None]
        return t.sub_(mean).div_(std)


class Resize:
    def __init__(self, size, interpolation=None):
        self.size, self.interpolation = size, interpolation

    def __call__(self, x):
        raise NotImplementedError("torchvision.transforms.Resize is outside the shim (600p DAVIS mode)")

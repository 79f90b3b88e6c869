"""File decoding shared by the test-time loaders: JPEG frames -> uint8 [T,H,W,3], palette PNGs -> uint8 [H,W] label maps.
PIL does the (CPU) entropy decoding like the reference's loaders; everything after it - normalisation, bicubic resize, one-hot,
padding - runs in HIP kernels (mivos_amd/clip_io.py) when a GPU `device` is given, or with the reference's own host arithmetic
when the loader runs inside a DataLoader worker process (which cannot touch the GPU: eval_interactive_davis.py:44 uses
num_workers=2)."""
import numpy as np
import torch
from PIL import Image

from .range_transform import IM_MEAN, IM_STD


def read_rgb(path):
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)


def read_labels(path):
    return np.array(Image.open(path).convert("P"), dtype=np.uint8)


def normalise_host(frames_u8):
    """uint8 [T,H,W,3] -> float32 [T,3,H,W]: ToTensor (x / 255) then Normalize ((x - mean) / std), the same fp32 operations in
    the same order as the reference's transform chain (davis_test_dataset.py:49-53), so the values are bit-identical."""
    x = torch.from_numpy(np.ascontiguousarray(frames_u8)).permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
    mean = torch.tensor(IM_MEAN, dtype=torch.float32)[None, :, None, None]
    std = torch.tensor(IM_STD, dtype=torch.float32)[None, :, None, None]
    return x.sub_(mean).div_(std)


def on_gpu(device):
    """True when `device` names a GPU and this process may use it (not a DataLoader worker)."""
    if device is None or torch.device(device).type != "cuda":
        return False
    from torch.utils.data import get_worker_info
    if get_worker_info() is not None:
        raise RuntimeError("a loader with device='cuda' cannot run inside DataLoader worker processes (num_workers > 0): "
                           "construct it with device=None there, or use num_workers=0")
    return True

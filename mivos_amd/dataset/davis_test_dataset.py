"""DAVISTestDataset for the MI355X engine - same constructor, attributes and `__getitem__` dictionary as the reference's
`dataset/davis_test_dataset.py:18-110` (`{'rgb' [T,3,H,W], 'gt' [K,T,1,H,W], 'info': name / num_frames / size_480p / labels}`),
so `eval_interactive_davis.py:41-52` and `generate_fusion.py` drive it unchanged.

``device=None`` (the reference's signature): CPU tensors, produced with the reference's host arithmetic - the form a
``DataLoader(num_workers=2)`` worker can deliver.  ``device='cuda:0'``: the decoded uint8 frames are uploaded (3 bytes per pixel
instead of 12) and normalised / one-hot encoded by HIP kernels (clip_io.ingest_frames / onehot_mask); 'rgb' and 'gt' then live
on the GPU, ready for InferenceCore (which keeps them resident), bit-identical to the host path.

Any other `resolution` (the reference's "600p" mode, davis_test_dataset.py:54-63: frames and annotations are read from
``JPEGImages/<resolution>`` / ``Annotations/<resolution>`` and `torchvision.transforms.Resize(600)` brings the short side to 600 -
bicubic on the normalised frames, nearest on the one-hot masks): the same two resizes, on the host with torch's bicubic / nearest
filters (what torchvision 0.8's tensor path calls), on the GPU with the HIP kernels the YouTube-VOS loader uses
(clip_io.ingest_frames(resize_to=...) within 2e-5 of the host filter, masks identical).  torchvision is not installed in this image, so
this mode is pinned by its restated size rule and the shared resize kernels, not by a golden vector of the reference."""
import os
from os import path

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data.dataset import Dataset

from ._decode import normalise_host, on_gpu, read_labels, read_rgb
from .onehot_util import all_to_onehot


class DAVISTestDataset(Dataset):
    def __init__(self, root, imset="2017/val.txt", resolution="480p", single_object=False, target_name=None, device=None):
        self.root, self.resolution, self.device = root, resolution, device
        self.mask_dir = path.join(root, "Annotations", resolution)
        self.mask480_dir = path.join(root, "Annotations", "480p")
        self.image_dir = path.join(root, "JPEGImages", resolution)
        self.videos, self.num_frames, self.num_objects, self.shape, self.size_480p = [], {}, {}, {}, {}
        with open(path.join(root, "ImageSets", imset), "r") as lines:
            for line in lines:
                video = line.rstrip("\n")
                if not video or (target_name is not None and target_name != video):
                    continue
                self.videos.append(video)
                self.num_frames[video] = len(os.listdir(path.join(self.image_dir, video)))
                first = read_labels(path.join(self.mask_dir, video, "00000.png"))
                self.num_objects[video] = np.max(first)
                self.shape[video] = np.shape(first)
                self.size_480p[video] = np.shape(read_labels(path.join(self.mask480_dir, video, "00000.png")))
        self.single_object = single_object

    @staticmethod
    def resized_size(h, w, size=600):
        """torchvision.transforms.Resize(size) with an int: the SHORT side becomes `size`, the other one int(size * long / short);
        unchanged when the short side already has that length."""
        if (w <= h and w == size) or (h <= w and h == size):
            return h, w
        return (int(size * h / w), size) if w < h else (size, int(size * w / h))

    def __len__(self):
        return len(self.videos)

    def __getitem__(self, index):
        video = self.videos[index]
        info = {"name": video, "num_frames": self.num_frames[video], "size_480p": self.size_480p[video]}
        frames, masks = [], []
        for f in range(self.num_frames[video]):
            frames.append(read_rgb(path.join(self.image_dir, video, "{:05d}.jpg".format(f))))
            mask_file = path.join(self.mask_dir, video, "{:05d}.png".format(f))
            masks.append(read_labels(mask_file) if path.exists(mask_file) else np.zeros_like(masks[0]))      # test-dev: first frame only
        frames, masks = np.stack(frames, 0), np.stack(masks, 0)
        if self.single_object:
            labels = [1]
            masks = (masks > 0.5).astype(np.uint8)
        else:
            labels = np.unique(masks[0])
            labels = labels[labels != 0]
        new_size = None if self.resolution == "480p" else self.resized_size(*masks.shape[-2:])
        if on_gpu(self.device):
            from .. import clip_io
            images = clip_io.ingest_frames(frames, self.device, resize_to=new_size, padded=False)[0]         # [T,3,H',W'] on the GPU
            gt = clip_io.onehot_masks(masks, labels, self.device, resize_to=new_size)[1:]                     # [K,T,1,H',W']: one upload
        else:
            images = normalise_host(frames)
            gt = torch.from_numpy(all_to_onehot(masks, labels)).float()
            if new_size is not None:                                  # davis_test_dataset.py:54-63, 98-99
                images = F.interpolate(images, size=new_size, mode="bicubic", align_corners=False)
                gt = F.interpolate(gt, size=new_size, mode="nearest")
            gt = gt.unsqueeze(2)
        info["labels"] = labels
        return {"rgb": images, "gt": gt, "info": info}

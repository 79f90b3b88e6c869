"""DAVISTestDataset for the MI355X engine - same constructor, attributes and `__getitem__` dictionary as the reference's
`dataset/davis_test_dataset.py:18-110` (`{'rgb' [T,3,H,W], 'gt' [K,T,1,H,W], 'info': name / num_frames / size_480p / labels}`),
so `eval_interactive_davis.py:41-52` and `generate_fusion.py` drive it unchanged.

``device=None`` (the reference's signature): CPU tensors, produced with the reference's host arithmetic - the form a
``DataLoader(num_workers=2)`` worker can deliver.  ``device='cuda:0'``: the decoded uint8 frames are uploaded (3 bytes per pixel
instead of 12) and normalised / one-hot encoded by HIP kernels (clip_io.ingest_frames / onehot_mask); 'rgb' and 'gt' then live
on the GPU, ready for InferenceCore (which keeps them resident), bit-identical to the host path.  Only the 480p mode exists here
(the reference's 600p mode resizes with torchvision.transforms.Resize; no caller on the path uses it)."""
import os
from os import path

import numpy as np
import torch
from torch.utils.data.dataset import Dataset

from ._decode import normalise_host, on_gpu, read_labels, read_rgb
from .onehot_util import all_to_onehot


class DAVISTestDataset(Dataset):
    def __init__(self, root, imset="2017/val.txt", resolution="480p", single_object=False, target_name=None, device=None):
        if resolution != "480p":
            raise NotImplementedError("mivos_amd.dataset.DAVISTestDataset: only resolution='480p' (the evaluation scripts' mode)")
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
        if on_gpu(self.device):
            from .. import clip_io
            images = clip_io.ingest_frames(frames, self.device, padded=False)[0]                              # [T,3,H,W] on the GPU
            gt = torch.stack([clip_io.onehot_mask(m, labels, self.device)[1:, 0] for m in masks], 1)           # [K,T,H,W]
        else:
            images = normalise_host(frames)
            gt = torch.from_numpy(all_to_onehot(masks, labels)).float()
        info["labels"] = labels
        return {"rgb": images, "gt": gt.unsqueeze(2), "info": info}

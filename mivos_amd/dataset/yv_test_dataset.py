"""YouTubeVOSTestDataset for the MI355X engine - same constructor and `__getitem__` dictionary as the reference's
`dataset/yv_test_dataset.py:16-119`: frames and masks of a video resized so that the short side is 480 (bicubic / nearest,
:102-109), `info` with name / num_objects / frames / size / gt_obj / label_convert / label_backward / labels.

``device=None``: the reference's host arithmetic (DataLoader workers).  ``device='cuda:0'``: uint8 upload, normalisation and the
bicubic resize in HIP kernels (clip_io.ingest_frames(resize_to=...): within 2e-5 of torch's CPU bicubic filter), nearest-neighbour
one-hot masks by clip_io.onehot_mask."""
import os
from os import path

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data.dataset import Dataset

from ._decode import normalise_host, on_gpu, read_labels, read_rgb


class YouTubeVOSTestDataset(Dataset):
    def __init__(self, data_root, split, device=None):
        self.image_dir = path.join(data_root, "vos", "all_frames", split, "JPEGImages")
        self.mask_dir = path.join(data_root, "vos", split, "Annotations")
        self.device = device
        self.videos, self.shape, self.frames = [], {}, {}
        for vid in sorted(os.listdir(self.image_dir)):
            self.frames[vid] = sorted(os.listdir(path.join(self.image_dir, vid)))
            self.videos.append(vid)
            first_mask = os.listdir(path.join(self.mask_dir, vid))[0]
            self.shape[vid] = np.shape(read_labels(path.join(self.mask_dir, vid, first_mask)))

    def To_onehot(self, mask, labels):
        return np.stack([(mask == l).astype(np.uint8) for l in labels], 0) if len(labels) else np.zeros((0,) + mask.shape, np.uint8)

    def All_to_onehot(self, masks, labels):
        out = np.zeros((len(labels),) + tuple(masks.shape), dtype=np.uint8)
        for n in range(masks.shape[0]):
            out[:, n] = self.To_onehot(masks[n], labels)
        return out

    def __len__(self):
        return len(self.videos)

    def __getitem__(self, idx):
        video = self.videos[idx]
        info = {"name": video, "num_objects": 0, "frames": self.frames[video], "size": self.shape[video], "gt_obj": {}}
        frames, masks = [], []
        for i, f in enumerate(self.frames[video]):
            frames.append(read_rgb(path.join(self.image_dir, video, f)))
            mask_file = path.join(self.mask_dir, video, f.replace(".jpg", ".png"))
            if path.exists(mask_file):
                masks.append(read_labels(mask_file))
                this_labels = np.unique(masks[-1])
                info["gt_obj"][i] = this_labels[this_labels != 0]
            else:
                masks.append(np.zeros(self.shape[video]))          # no annotation -> nothing in it (float64 zeros, like the reference)
        frames, masks = np.stack(frames, 0), np.stack(masks, 0)
        labels = np.unique(masks).astype(np.uint8)
        labels = labels[labels != 0]
        info["label_convert"], info["label_backward"] = {}, {}
        for n, l in enumerate(labels, start=1):
            info["label_convert"][l] = n
            info["label_backward"][n] = l
        h, w = masks.shape[-2:]
        new_size = (h * 480 // w, 480) if h > w else (480, w * 480 // h)
        if on_gpu(self.device):
            from .. import clip_io
            images = clip_io.ingest_frames(frames, self.device, resize_to=new_size, padded=False)[0]
            gt = clip_io.onehot_masks(masks.astype(np.uint8), labels, self.device, resize_to=new_size)[1:]       # [K,T,1,H',W']
        else:
            images = F.interpolate(normalise_host(frames), size=new_size, mode="bicubic", align_corners=False)
            gt = torch.from_numpy(self.All_to_onehot(masks, labels)).float().unsqueeze(2)
            gt = F.interpolate(gt, size=(1, *new_size), mode="nearest")
        info["labels"] = labels
        return {"rgb": images, "gt": gt, "info": info}

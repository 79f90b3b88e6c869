"""Golden vectors of the test-time loaders (reference `dataset/davis_test_dataset.py:18-110`, `dataset/yv_test_dataset.py:16-119`),
produced by the UNMODIFIED reference classes on the committed mini-datasets (oracle/make_mini_dataset.py).

TEST INFRASTRUCTURE ONLY (this container; torchvision.transforms comes from oracle/ref_shim).  Writes
``tests/golden/dataset_small.npz``:
  davis_<seq>_rgb / _gt / _labels / _size480   the full `__getitem__` tensors of both mini-DAVIS sequences (gt as uint8)
  yv_rgb_sub / yv_rgb_sums                     every 6th pixel of the resized 480 x 768 frames + per-frame float64 (sum, sum of
                                               squares) of ALL pixels (the full tensor is 13 MB)
  yv_gt_bits / yv_gt_shape / yv_labels / yv_gt_obj / yv_size

    python -m oracle.make_mini_dataset && python -m oracle.make_golden_dataset
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
YV_SUB = 6


def main():
    ref_loader.load_reference()
    ddav = importlib.import_module("dataset.davis_test_dataset")
    dyv = importlib.import_module("dataset.yv_test_dataset")
    assert ddav.__file__.startswith(ref_loader.REFERENCE_ROOT) and dyv.__file__.startswith(ref_loader.REFERENCE_ROOT)
    out = {}
    ds = ddav.DAVISTestDataset(os.path.join(G, "mini_davis", "trainval"), imset="2017/val.txt")
    names = []
    for i in range(len(ds)):
        d = ds[i]
        n = d["info"]["name"]
        names.append(n)
        out[f"davis_{n}_rgb"] = d["rgb"].numpy()
        out[f"davis_{n}_gt"] = d["gt"].numpy().astype(np.uint8)
        out[f"davis_{n}_labels"] = np.asarray(d["info"]["labels"])
        out[f"davis_{n}_size480"] = np.asarray(d["info"]["size_480p"])
        print("davis", n, tuple(d["rgb"].shape), tuple(d["gt"].shape), d["info"]["labels"], d["info"]["num_frames"])
    out["davis_names"] = np.asarray(names)
    so = ddav.DAVISTestDataset(os.path.join(G, "mini_davis", "trainval"), imset="2017/val.txt", single_object=True, target_name="blackswan")
    d = so[0]
    out["davis_single_gt"] = d["gt"].numpy().astype(np.uint8)
    yv = dyv.YouTubeVOSTestDataset(os.path.join(G, "mini_yv"), "valid")
    d = yv[0]
    rgb, gt = d["rgb"], d["gt"]
    out["yv_rgb_sub"] = rgb[..., ::YV_SUB, ::YV_SUB].numpy()
    out["yv_rgb_sums"] = np.stack([rgb.double().sum(dim=(1, 2, 3)).numpy(), (rgb.double() ** 2).sum(dim=(1, 2, 3)).numpy()])
    out["yv_gt_bits"] = np.packbits(gt.numpy().astype(np.uint8))
    out["yv_gt_shape"] = np.asarray(gt.shape)
    out["yv_labels"] = np.asarray(d["info"]["labels"])
    out["yv_size"] = np.asarray(d["info"]["size"])
    out["yv_info"] = json.dumps(dict(name=d["info"]["name"], frames=d["info"]["frames"], gt_obj={str(k): [int(x) for x in v] for k, v in d["info"]["gt_obj"].items()},
                                     label_convert={str(int(k)): int(v) for k, v in d["info"]["label_convert"].items()},
                                     label_backward={str(int(k)): int(v) for k, v in d["info"]["label_backward"].items()}))
    print("yv", d["info"]["name"], tuple(rgb.shape), tuple(gt.shape), d["info"]["labels"], d["info"]["gt_obj"])
    path = os.path.join(G, "dataset_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

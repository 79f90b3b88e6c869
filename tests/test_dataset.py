"""Test-time loaders: mivos_amd/dataset/{davis,yv}_test_dataset.py against the outputs of the UNMODIFIED
reference classes (dataset/davis_test_dataset.py:18-110, dataset/yv_test_dataset.py:16-119) on the committed mini-datasets
(tests/golden/mini_davis, mini_yv; fixture tests/golden/dataset_small.npz, oracle/make_golden_dataset.py).  Host path (what a
DataLoader worker delivers) on CPU; the HIP ingest path with -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

from mivos_amd.dataset.davis_test_dataset import DAVISTestDataset
from mivos_amd.dataset.yv_test_dataset import YouTubeVOSTestDataset

YV_SUB = 6


@pytest.fixture(scope="module")
def g(golden_dir):
    with np.load(os.path.join(golden_dir, "dataset_small.npz")) as z:
        return {k: z[k] for k in z.files}


def _check_davis(ds, g, exact_rgb=True):
    assert len(ds) == 2 and ds.videos == list(g["davis_names"])
    for i, name in enumerate(ds.videos):
        d = ds[i]
        rgb, gt = d["rgb"].cpu(), d["gt"].cpu()
        assert rgb.dtype == torch.float32 and gt.dtype == torch.float32
        assert rgb.shape == g[f"davis_{name}_rgb"].shape and torch.equal(rgb, torch.from_numpy(g[f"davis_{name}_rgb"]))     # bit-identical
        assert gt.shape == g[f"davis_{name}_gt"].shape and torch.equal(gt, torch.from_numpy(g[f"davis_{name}_gt"]).float())
        info = d["info"]
        assert info["name"] == name and info["num_frames"] == rgb.shape[0] and tuple(info["size_480p"]) == tuple(g[f"davis_{name}_size480"])
        assert np.array_equal(info["labels"], g[f"davis_{name}_labels"]) and ds.num_objects[name] == len(info["labels"])


def _check_yv(ds, g, rgb_tol):
    assert len(ds) == 1
    d = ds[0]
    rgb, gt = d["rgb"].cpu(), d["gt"].cpu()
    assert tuple(gt.shape) == tuple(g["yv_gt_shape"]) and rgb.shape == (3, 3, 480, 768)
    assert float((rgb[..., ::YV_SUB, ::YV_SUB] - torch.from_numpy(g["yv_rgb_sub"])).abs().max()) <= rgb_tol
    sums = np.stack([rgb.double().sum(dim=(1, 2, 3)).numpy(), (rgb.double() ** 2).sum(dim=(1, 2, 3)).numpy()])
    assert np.allclose(sums, g["yv_rgb_sums"], rtol=0 if rgb_tol == 0 else 1e-5, atol=0 if rgb_tol == 0 else 1.0)
    want = np.unpackbits(g["yv_gt_bits"])[:gt.numel()].reshape(gt.shape)
    assert torch.equal(gt, torch.from_numpy(want).float())
    info, gi = d["info"], json.loads(str(g["yv_info"]))
    assert info["name"] == gi["name"] and info["frames"] == gi["frames"] and tuple(info["size"]) == tuple(g["yv_size"])
    assert np.array_equal(info["labels"], g["yv_labels"]) and info["num_objects"] == 0
    assert {str(k): [int(x) for x in v] for k, v in info["gt_obj"].items()} == gi["gt_obj"]
    assert {str(int(k)): int(v) for k, v in info["label_convert"].items()} == gi["label_convert"]
    assert {str(int(k)): int(v) for k, v in info["label_backward"].items()} == gi["label_backward"]


def test_davis_loader_matches_the_reference_bitwise(golden_dir, g):
    ds = DAVISTestDataset(os.path.join(golden_dir, "mini_davis", "trainval"), imset="2017/val.txt")
    _check_davis(ds, g)
    so = DAVISTestDataset(os.path.join(golden_dir, "mini_davis", "trainval"), single_object=True, target_name="blackswan")
    assert len(so) == 1 and torch.equal(so[0]["gt"], torch.from_numpy(g["davis_single_gt"]).float()) and so[0]["info"]["labels"] == [1]
    with pytest.raises(NotImplementedError):
        DAVISTestDataset(os.path.join(golden_dir, "mini_davis", "trainval"), resolution="1080p")


def test_yv_loader_matches_the_reference_bitwise(golden_dir, g):
    _check_yv(YouTubeVOSTestDataset(os.path.join(golden_dir, "mini_yv"), "valid"), g, rgb_tol=0.0)


def test_davis_loader_through_a_dataloader_like_the_entry_script(golden_dir, g):
    """eval_interactive_davis.py:43-52: DataLoader(batch_size=1, num_workers=2), then data['rgb'], len(data['info']['labels'][0]),
    data['info']['name'][0]."""
    from torch.utils.data import DataLoader
    ds = DAVISTestDataset(os.path.join(golden_dir, "mini_davis", "trainval"), imset="2017/val.txt")
    seen = {}
    for data in DataLoader(ds, batch_size=1, shuffle=False, num_workers=2):
        seen[data["info"]["name"][0]] = (data["rgb"], len(data["info"]["labels"][0]))
    assert list(seen) == ["blackswan", "seqb"] and seen["blackswan"][1] == 2 and seen["seqb"][1] == 1
    assert seen["blackswan"][0].shape == (1, 5, 3, 128, 157) and torch.equal(seen["blackswan"][0][0], torch.from_numpy(g["davis_blackswan_rgb"]))


@pytest.mark.gpu
def test_loaders_on_the_gpu_ingest_path(golden_dir, g):
    """device='cuda:0': uint8 upload + HIP normalise / bicubic / one-hot kernels.  DAVIS: bit-identical to the reference's loader;
    YouTube-VOS: the bicubic filter within 2e-5 of torch's CPU one, masks exact."""
    ds = DAVISTestDataset(os.path.join(golden_dir, "mini_davis", "trainval"), imset="2017/val.txt", device="cuda:0")
    assert ds[0]["rgb"].is_cuda and ds[0]["gt"].is_cuda
    _check_davis(ds, g)
    _check_yv(YouTubeVOSTestDataset(os.path.join(golden_dir, "mini_yv"), "valid", device="cuda:0"), g, rgb_tol=2e-5)

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


def _full_resolution_root(golden_dir, tmp_path):
    """A DAVIS root whose `Full-Resolution` trees are the mini dataset's 480p trees (symlinks): what the loader's non-480p mode reads."""
    src = os.path.join(golden_dir, "mini_davis", "trainval")
    root = tmp_path / "davis"
    for sub in ("JPEGImages", "Annotations"):
        os.makedirs(root / sub)
        os.symlink(os.path.join(src, sub, "480p"), root / sub / "480p")
        os.symlink(os.path.join(src, sub, "480p"), root / sub / "Full-Resolution")
    os.symlink(os.path.join(src, "ImageSets"), root / "ImageSets")
    return str(root)


def test_davis_loader_resize_mode(golden_dir, tmp_path):
    """davis_test_dataset.py:54-63, 98-99: any resolution but '480p' reads JPEGImages/<resolution> and brings the SHORT side to 600
    (torchvision Resize(600): bicubic on the normalised frames, nearest on the one-hot masks)."""
    rs = DAVISTestDataset.resized_size
    assert rs(480, 854) == (600, 1067) and rs(1080, 1920) == (600, 1066) and rs(854, 480) == (1067, 600) and rs(600, 800) == (600, 800) and rs(128, 157) == (600, 735)
    root = _full_resolution_root(golden_dir, tmp_path)
    base = DAVISTestDataset(root, imset="2017/val.txt")[0]
    big = DAVISTestDataset(root, imset="2017/val.txt", resolution="Full-Resolution")[0]
    assert big["rgb"].shape == (5, 3, 600, 735) and big["gt"].shape == (2, 5, 1, 600, 735) and big["info"]["size_480p"] == base["info"]["size_480p"]
    assert torch.equal(big["rgb"], torch.nn.functional.interpolate(base["rgb"], size=(600, 735), mode="bicubic", align_corners=False))
    yy = (torch.arange(600) * (128 / 600)).floor().long().clamp(max=127)          # nearest: src = floor(dst * in / out)
    xx = (torch.arange(735) * (157 / 735)).floor().long().clamp(max=156)
    assert torch.equal(big["gt"][:, :, 0], base["gt"][:, :, 0][:, :, yy][:, :, :, xx])
    assert set(big["gt"].unique().tolist()) <= {0.0, 1.0}


@pytest.mark.gpu
def test_davis_loader_resize_mode_on_the_gpu(golden_dir, tmp_path):
    root = _full_resolution_root(golden_dir, tmp_path)
    host = DAVISTestDataset(root, imset="2017/val.txt", resolution="Full-Resolution")[1]
    dev = DAVISTestDataset(root, imset="2017/val.txt", resolution="Full-Resolution", device="cuda:0")[1]
    assert dev["rgb"].is_cuda and dev["gt"].is_cuda and dev["rgb"].shape == host["rgb"].shape and dev["gt"].shape == host["gt"].shape
    assert float((dev["rgb"].cpu() - host["rgb"]).abs().max()) < 5e-5 and torch.equal(dev["gt"].cpu(), host["gt"])


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


def test_suite_runner_over_the_loaders(golden_dir, tmp_path):
    """eval_suite.dataset_suite / dataset_factory / png_writer: the sharded suite loop (BASELINE config 4's protocol: first-frame annotation, propagate
    to the end) over the real loaders, with a stand-in engine on CPU: specs from metadata only, first-frame masks one-hot with background, one
    directory of palette PNGs per video."""
    from PIL import Image
    from mivos_amd import eval_suite as ES

    class StubCore:
        def __init__(self, prop, fuse, images, k, mem_profile=0, mem_freq=5, device="cpu"):
            self.t, self.k = images.shape[1], k
            assert images.dim() == 5 and images.shape[0] == 1 and images.shape[2] == 3

        def interact(self, mask, idx):
            lab = mask[:, 0].argmax(0).numpy().astype(np.uint8)
            return np.stack([np.roll(lab, t, axis=1) for t in range(self.t)], 0)

    dav = DAVISTestDataset(os.path.join(golden_dir, "mini_davis", "trainval"), imset="2017/val.txt")
    specs = ES.dataset_suite(dav)
    assert [(s.clip_id, s.frames, s.objects, s.height, s.width) for s in specs] == [(0, 5, 2, 128, 157), (1, 4, 1, 128, 157)]
    m = ES.first_frame_mask(dav[0]["gt"])
    assert m.shape == (3, 1, 128, 157) and torch.equal(m.sum(0), torch.ones(1, 128, 157)) and float(m[0].mean()) > 0.3
    palette = Image.open(os.path.join(golden_dir, "mini_davis", "trainval", "Annotations", "480p", "blackswan", "00000.png")).getpalette()
    recs = ES.run_suite(specs, ES.dataset_factory(dav, None, None, device="cpu", core_cls=StubCore), on_clip=ES.png_writer(dav, str(tmp_path), palette))
    assert ES.summarize(recs, 2)["frames"] == 4 + 3
    first = np.array(Image.open(str(tmp_path / "blackswan" / "00000.png")))
    want = np.array(Image.open(os.path.join(golden_dir, "mini_davis", "trainval", "Annotations", "480p", "blackswan", "00000.png")))
    assert np.array_equal(first, want) and sorted(os.listdir(str(tmp_path / "seqb"))) == [f"{i:05d}.png" for i in range(4)]
    yv = YouTubeVOSTestDataset(os.path.join(golden_dir, "mini_yv"), "valid")
    ys = ES.dataset_suite(yv)
    assert [(s.frames, s.height, s.width) for s in ys] == [(3, 480, 768)]
    r2 = ES.run_suite(ys, ES.dataset_factory(yv, None, None, device="cpu", core_cls=StubCore))
    assert r2[0]["frames"] == 2 and r2[0]["objects"] == 1


@pytest.mark.gpu
def test_suite_runner_over_the_loaders_on_the_engine(golden_dir, synthetic_states):
    """The same loop with the real engine: mini-DAVIS through DAVISTestDataset(device='cuda:0') -> InferenceCore -> interact(first-frame annotation)."""
    from mivos_amd import eval_suite as ES
    from mivos_amd.model.fusion_net import FusionNet
    from mivos_amd.model.propagation.prop_net import PropagationNetwork
    prop, fuse = PropagationNetwork(top_k=50), FusionNet()
    prop.load_state_dict(synthetic_states[0])
    fuse.load_state_dict(synthetic_states[1])
    prop, fuse = prop.to("cuda:0").eval(), fuse.to("cuda:0").eval()
    dav = DAVISTestDataset(os.path.join(golden_dir, "mini_davis", "trainval"), imset="2017/val.txt", device="cuda:0")
    got = {}
    recs = ES.run_suite(ES.dataset_suite(dav), ES.dataset_factory(dav, prop, fuse, device="cuda:0"), sync=torch.cuda.synchronize,
                        on_clip=lambda spec, masks: got.__setitem__(spec.clip_id, masks))
    assert ES.summarize(recs, 2)["frames"] == 7
    for i, (t, k) in enumerate(((5, 2), (4, 1))):
        m = got[i]
        assert m.shape == (t, 128, 157) and m.dtype == np.uint8 and set(np.unique(m)) <= set(range(k + 1))
        want = dav[i]["gt"][:, 0, 0].cpu()                                      # frame 0 is the annotation itself
        assert all(np.array_equal(m[0] == j + 1, want[j].numpy() > 0.5) for j in range(k))

"""Ingest / egress either side of the path (mivos_amd/clip_io.py): PNG writer on CPU, HIP ingest kernels on the GPU against
the reference loaders' torch semantics (dataset/davis_test_dataset.py, dataset/yv_test_dataset.py)."""
import io

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mivos_amd import clip_io


def test_palette_png_roundtrip(tmp_path):
    from PIL import Image
    r = np.random.RandomState(0)
    masks = (r.rand(3, 37, 53) * 6).astype(np.uint8)
    palette = [int(v) for v in r.randint(0, 256, 768)]
    clip_io.write_palette_png(masks, palette, str(tmp_path))
    for i in range(3):
        im = Image.open(str(tmp_path / f"{i:05d}.png"))
        assert im.mode == "P" and np.array_equal(np.array(im), masks[i]) and im.getpalette()[:768] == palette
    im = Image.open(io.BytesIO(clip_io.encode_palette_png(masks[0], palette[:30])))     # short palette
    assert np.array_equal(np.array(im), masks[0])


def test_geometry_helpers():
    assert clip_io._pad16(480, 854) == (480, 864, (5, 5, 0, 0)) and clip_io._pad16(101, 35) == (112, 48, (6, 7, 5, 6))
    assert clip_io.yv_480p_size(720, 1280) == (480, 853) and clip_io.yv_480p_size(1280, 720) == (853, 480)


@pytest.mark.gpu
def test_ingest_matches_the_reference_loader_semantics():
    from mivos_amd.util.tensor_util import pad_divide_by
    r = np.random.RandomState(1)
    frames = r.randint(0, 256, (3, 90, 130, 3)).astype(np.uint8)
    mean, std = torch.tensor(clip_io.IM_MEAN)[None, :, None, None], torch.tensor(clip_io.IM_STD)[None, :, None, None]
    ref = (torch.from_numpy(frames).permute(0, 3, 1, 2).float().div(255) - mean) / std      # ToTensor + Normalize
    got, pad = clip_io.ingest_frames(frames)
    want, wpad = pad_divide_by(ref[None], 16)
    assert pad == wpad and got.shape == want.shape and torch.equal(got.cpu(), want)         # bit-identical, padding included
    # YouTube-VOS path: bicubic to the 480p rule, then pad
    size = clip_io.yv_480p_size(90, 130)
    refr = F.interpolate(ref, size=size, mode="bicubic", align_corners=False)
    gotr = clip_io.ingest_frames(frames, resize_to=size, padded=False)
    assert gotr.shape == (1, 3, 3) + size and float((gotr.cpu()[0] - refr).abs().max()) < 2e-5
    # one-hot ground truth of the annotated frame (+ nearest resize)
    lab = (r.rand(90, 130) * 4).astype(np.uint8) * 3                       # labels 0, 3, 6, 9
    oh = clip_io.onehot_mask(lab, [3, 6, 9]).cpu()
    assert oh.shape == (4, 1, 90, 130) and torch.equal(oh[1:, 0], torch.stack([torch.from_numpy(lab == v).float() for v in (3, 6, 9)]))
    assert torch.equal(oh.sum(0), torch.ones(1, 90, 130))
    ohr = clip_io.onehot_mask(lab, [3, 6, 9], resize_to=size).cpu()
    want = F.interpolate(torch.stack([torch.from_numpy(lab == v).float() for v in (3, 6, 9)])[None, :, None], size=(1,) + size, mode="nearest")[0, :, 0]
    assert torch.equal(ohr[1:, 0], want)

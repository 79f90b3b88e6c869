"""Scribble-to-mask network (mivos_amd/model/s2m) and the DAVISProcessor junction (mivos_amd/davis_processor.py) against
the reference's golden vector and the CPU oracle (needs an MI355X)."""
import os

import numpy as np
import pytest
import torch

from mivos_amd.model.s2m.s2m_network import deeplabv3plus_resnet50
from oracle import s2m_oracle as SO
from oracle import stm_oracle as O
from oracle import weights as Wt

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda:0"


@pytest.fixture(scope="module")
def s2m():
    net = deeplabv3plus_resnet50()
    net.load_state_dict(Wt.make_s2m_state(0))
    return net.to(DEV).eval()


def test_s2m_golden(s2m, golden_dir):
    with np.load(os.path.join(golden_dir, "s2m_small.npz")) as z:
        x, ref = torch.from_numpy(z["s2m_in"]), torch.from_numpy(z["s2m_out"])
    got = s2m(x.to(DEV)).cpu()
    d = float((got - ref).abs().max())
    print(f"S2M golden: max|dlogit| {d:.2e}, range [{float(ref.min()):.1f}, {float(ref.max()):.1f}]")
    assert got.shape == ref.shape and d < 1e-3


def test_s2m_480p_and_dilated_layers_vs_oracle(s2m):
    """480x864 (30x54 stride-16 grid: the atrous rates 6/12/18 reach beyond the map on one axis) and a batch of 2."""
    sd = Wt.make_s2m_state(0)
    images, gt = O.synthetic_clip(2, 480, 864, 2, seed=9)
    g = torch.Generator().manual_seed(2)
    pos = (torch.rand(2, 1, 480, 864, generator=g) > 0.97).float() * gt[:, 1]
    neg = (torch.rand(2, 1, 480, 864, generator=g) > 0.97).float() * (1 - gt[:, 1])
    x = torch.cat([images[0], gt[:, 1], pos, neg], 1)
    ref = SO.s2m_forward(sd, x)
    got = s2m(x.to(DEV)).cpu()
    d = (got - ref).abs()
    dp = float((torch.sigmoid(got) - torch.sigmoid(ref)).abs().max())
    print(f"S2M 480p: max|dlogit| {float(d.max()):.2e}, range [{float(ref.min()):.1f}, {float(ref.max()):.1f}], max|dprob| {dp:.2e}")
    # 53 convolutions deep with logits of +-26 (synthetic weights): fp32 rounding-level agreement relative to the value range,
    # and the quantity that is used downstream - the sigmoid probability - within the 1e-3-logit bar's equivalent
    assert got.shape == (2, 1, 480, 864) and float(d.max()) < 1e-4 * float(ref.abs().max()) and dp < 5e-4


def test_davis_processor_schedule_and_to_mask_vs_oracle(s2m, synthetic_states):
    """The DAVIS schedule (interactions 0, 1 only update the mask, the third propagates; davis_processor.py:72-82) and the
    scribble -> S2M -> hard aggregate path vs the oracle."""
    from mivos_amd.davis_processor import DAVISProcessor
    from mivos_amd.model.fusion_net import FusionNet
    from mivos_amd.model.propagation.prop_net import PropagationNetwork
    sd, fsd = synthetic_states
    prop, fuse = PropagationNetwork(top_k=20), FusionNet()
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    K = 2
    images, gt = O.synthetic_clip(4, 100, 141, K, seed=12)               # not multiples of 16
    proc = DAVISProcessor(prop, fuse, s2m, images, K, device=DEV)
    r = np.random.RandomState(3)
    lab = gt[1].argmax(0)[0].numpy()                                       # frame 1
    scr = np.full((100, 141), -1, np.int64)
    pick = r.rand(100, 141) > 0.985
    scr[pick] = lab[pick]
    mask = proc.mask_from_scribble_mask(scr, 1).cpu()
    img, _ = O.pad_divide_by(images, 16)
    ref = SO.to_mask(Wt.make_s2m_state(0), img[:, 1], torch.zeros(1, 112, 144, dtype=torch.uint8), torch.from_numpy(scr), K)
    assert mask.shape == ref.shape == (K + 1, 1, 112, 144)
    # hard aggregation (logits x 1000) saturates: compare the decision and the probabilities away from ties
    agree = (mask.argmax(0) == ref.argmax(0)).float().mean()
    print(f"to_mask: argmax agreement {float(agree):.6f}")
    assert float(agree) > 0.9995
    out, nxt, idx = proc.interact_scribble_mask(scr, 1)
    assert nxt == [1] and idx == 1 and out.shape == (4, 100, 141) and proc.processor.propagated_frames == 0
    out, nxt, _ = proc.interact_scribble_mask(scr, 1)
    assert nxt == [1] and proc.processor.propagated_frames == 0
    out, nxt, _ = proc.interact_scribble_mask(scr, 1)                      # third interaction: schedule[0] == 2 -> propagate
    assert nxt is None and proc.processor.propagated_frames == 3 and proc.davis_schedule == [5, 7]
    assert out.dtype == np.uint8 and out.shape == (4, 100, 141)

"""Calibration of the synthetic weights (TEST INFRASTRUCTURE; run once by oracle/make_golden.py).

The weight recipe itself (specs, seeded values, conditioning knobs) lives in
``mivos_amd/util/synthetic.py`` so that bench.py can build the same networks without touching the
oracle; everything is re-exported here for the tests.
"""
from collections import OrderedDict

import torch

from mivos_amd.util.synthetic import *          # noqa: F401,F403
from mivos_amd.util.synthetic import (KEY_STD, LOGIT_STD, _apply_calibration, make_fuse_state,  # noqa: F401
                                      make_prop_state, make_s2m_state, GOLDEN_DIR, state_fingerprint, prop_spec, fuse_spec, s2m_spec)


def calibrate(seed=0, size=(128, 160)):
    """One calibration pass over a synthetic clip, run ONCE in this container by
    oracle/make_golden.py; the result is committed so every machine uses identical
    numbers.  BatchNorm layers get 'measure batch statistics, store them as running stats,
    normalise with them'; the BN-free convs (KeyValue, decoder, FusionNet) are rescaled
    LSUV-style to unit output std so that keys, values and mask logits have the O(1)
    scale of a trained network (raw random init gives affinities of +-1000).
    Returns (prop_calib, fuse_calib): name -> tensor, gains under 'gain:<conv name>'."""
    from . import stm_oracle as O
    sd, fsd = make_prop_state(seed, calib=None), make_fuse_state(seed, calib=None)
    h, w = size
    images, gt = O.synthetic_clip(4, h, w, 2, seed=100 + seed)
    lsuv = {k[:-7]: 1.0 for k in sd if k.endswith(".weight") and (k.startswith("kv_") or k.startswith("decoder."))}
    lsuv["decoder.pred"] = LOGIT_STD
    lsuv["kv_m_f16.key_proj"] = lsuv["kv_q_f16.key_proj"] = KEY_STD
    pc = {"__lsuv__": lsuv}
    with O.bn_calibration(pc):
        # BN statistics over all 4 frames (batch), both encoders
        frames = images[0]
        O.rgb_encoder(sd, frames)
        O.mask_rgb_encoder(sd, frames, gt[:, 1], gt[:, 2])
        sd2 = OrderedDict(sd)
        _apply_calibration(sd2, {k: v for k, v in pc.items() if not k.startswith("__") and not k.startswith("gain:")})
    pc2 = {"__lsuv__": lsuv, "__bn__": False}
    with O.bn_calibration(pc2):
        # conv gains along the real data flow (each conv is rescaled on the fly, so one
        # pass is self-consistent): memorize frame 0, query frame 1, read + decode
        mk, mv = O.memorize(sd2, frames[0:1], gt[0, 1:])
        q = O.get_query_values(sd2, frames[1:2])
        logit = O.segment_logits(sd2, mk, mv, *q, top_k=20)
    prop_calib = {k: v for k, v in pc.items() if not k.startswith("__")}
    prop_calib.update({k: v for k, v in pc2.items() if k.startswith("gain:")})
    # BN stats measured in the 2nd pass would overwrite nothing we keep: filter them out
    prop_calib = {k: v for k, v in prop_calib.items() if k.startswith("gain:") or k in sd}
    # FusionNet: unit std after every conv, logit std 1.5
    fl = {k[:-7]: 1.0 for k in fsd if k.endswith(".weight")}
    fl["final_conv"] = LOGIT_STD
    fc = {"__lsuv__": fl}
    prob = torch.sigmoid(logit)
    attn = torch.rand(1, 2, h, w) * 0.3
    with O.bn_calibration(fc):
        O.fusion_net(fsd, frames[1:2], prob[0:1], gt[1, 1:2], attn, torch.tensor([[0.4, 0.6]]))
    fuse_calib = {k: v for k, v in fc.items() if k.startswith("gain:")}
    return prop_calib, fuse_calib


def calibrate_s2m(seed=0, size=(128, 160)):
    """BN running statistics of the synthetic S2M network (batch of 4 frames: image + current mask + positive / negative
    scribble planes) and an LSUV gain that gives its logit the std of a trained network's (LOGIT_STD)."""
    from . import s2m_oracle as SO
    from . import stm_oracle as O
    sd = make_s2m_state(seed, calib=None)
    h, w = size
    images, gt = O.synthetic_clip(4, h, w, 2, seed=200 + seed)
    g = torch.Generator().manual_seed(17 + seed)
    pos = (torch.rand(4, 1, h, w, generator=g) > 0.97).float() * gt[:, 1]
    neg = (torch.rand(4, 1, h, w, generator=g) > 0.97).float() * (1 - gt[:, 1])
    x = torch.cat([images[0], gt[:, 1] * (torch.rand(4, 1, h, w, generator=g) > 0.3).float(), pos, neg], 1)
    pc = {"__lsuv__": {"classifier.classifier.3": LOGIT_STD}}
    with O.bn_calibration(pc):
        SO.s2m_forward(sd, x)
    return {k: v for k, v in pc.items() if not k.startswith("__")}

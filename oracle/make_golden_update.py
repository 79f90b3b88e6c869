"""Golden vector of `InferenceCore.update_mask_only` (reference `inference_core.py:273-293`; the DAVIS schedule calls it on 5 of
the 8 interactions of a session, `davis_processor.py:75-82`), produced by the UNMODIFIED reference on PyTorch-CPU.

TEST INFRASTRUCTURE ONLY (this container).  Writes ``tests/golden/update_small.npz``: a 5-frame 100x141 clip (padded to 112x144,
pad = (1, 2, 6, 6): both axes padded unevenly) with 3 objects; after `interact(gt[0], 0)` three `update_mask_only` calls - soft
probabilities with exact ties between channels (argmax picks the first), a one-hot mask, and the frame's own propagated
probabilities - each followed by a copy of `np_masks` and of `masks[idx]`.

    python -m oracle.make_golden_update
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle import stm_oracle as O  # noqa: E402
from oracle import weights as Wt  # noqa: E402

CFG = dict(t=5, h=100, w=141, k=3, seed=17, mem_freq=2, top_k=20, calls=[1, 3, 2])


def update_inputs(cfg, prob_third):
    """The three prob_mask arguments [K+1,1,nh,nw] (padded size) of the golden session; `prob_third` = the reference core's
    probabilities of frame calls[2] after interact(gt[0], 0) (stored in the fixture as `input_2`).  Deterministic (numpy
    RandomState), shared with the tests."""
    nh, nw = prob_third.shape[-2:]
    r = np.random.RandomState(5)
    soft = torch.from_numpy(r.rand(cfg["k"] + 1, 1, nh, nw).astype(np.float32))
    soft[:, :, ::3, ::2] = 0.5                                        # exact ties across all channels: argmax -> channel 0
    soft[2, :, 1::3, 1::2] = soft[1, :, 1::3, 1::2]                   # ties between two objects: the lower index wins
    lab = torch.from_numpy(r.randint(0, cfg["k"] + 1, size=(nh, nw)))
    onehot = torch.stack([(lab == j).float() for j in range(cfg["k"] + 1)], 0).unsqueeze(1)
    return [soft, onehot, prob_third.clone()]


def main():
    torch.set_grad_enabled(False)
    ref, prop, fuse = ref_loader.build_reference_networks(top_k=CFG["top_k"])
    sd, fsd = Wt.make_prop_state(0), Wt.make_fuse_state(0)
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    images, gt = O.synthetic_clip(CFG["t"], CFG["h"], CFG["w"], CFG["k"], CFG["seed"])
    core = ref["inference_core"].InferenceCore(prop, fuse, images, CFG["k"], mem_profile=0, mem_freq=CFG["mem_freq"], device="cpu")
    out = {"config": json.dumps(CFG), "pad": np.asarray(core.pad)}
    out["masks_interact"] = core.interact(gt[0], 0).copy()
    out["input_2"] = core.prob[:, CFG["calls"][2]].numpy().copy()
    ocore = O.OracleCore(sd, fsd, images, CFG["k"], mem_freq=CFG["mem_freq"], top_k=CFG["top_k"])
    ocore.interact(gt[0], 0)
    for n, (idx, pm) in enumerate(zip(CFG["calls"], update_inputs(CFG, core.prob[:, CFG["calls"][2]]))):
        res = core.update_mask_only(pm, idx)
        ores = ocore.update_mask_only(pm, idx)
        out[f"np_masks_{n}"] = res.copy()
        out[f"masks_idx_{n}"] = core.masks[idx].numpy().copy()
        print(f"update {n}: idx {idx} -> {res.shape} {res.dtype}, labels {np.unique(res[idx]).tolist()}, oracle mismatch {int((res != ores).sum())}")
    path = os.path.join(ROOT, "tests", "golden", "update_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

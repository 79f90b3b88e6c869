"""Golden vectors of the fusion-data generator (reference `generation/fusion_generator.py:12-101`, driven like
`generate_fusion.py:68-120`), produced by the UNMODIFIED reference on PyTorch-CPU.

TEST INFRASTRUCTURE ONLY (this container).  Writes ``tests/golden/gen_small.npz``: a 6-frame 120x150 clip with 2 objects,
`FusionGenerator(prop_net, images, mem_freq=2)`, `reset(2)`, `interact_mask(mask, idx, left, right)` for two reference frames
(one with range limits inside the clip) -> the returned probabilities [K+1, T, H, W].

    python -m oracle.make_golden_gen
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
from oracle import stm_oracle as O  # noqa: E402
from oracle import weights as Wt  # noqa: E402

CFG = dict(t=6, h=120, w=150, k=2, seed=21, mem_freq=2, top_k=20, calls=[[2, 0, 5], [4, 1, 5]])   # (idx, left_limit, right_limit)


def main():
    torch.set_grad_enabled(False)
    ref, prop, _ = ref_loader.build_reference_networks(top_k=CFG["top_k"])
    prop.load_state_dict(Wt.make_prop_state(0))
    gen_mod = importlib.import_module("generation.fusion_generator")
    images, gt = O.synthetic_clip(CFG["t"], CFG["h"], CFG["w"], CFG["k"], CFG["seed"])
    out = {"config": json.dumps(CFG)}
    proc = gen_mod.FusionGenerator(prop, images, CFG["mem_freq"])
    for n, (idx, left, right) in enumerate(CFG["calls"]):
        proc.reset(CFG["k"])
        probs = proc.interact_mask(gt[idx, 1:], idx, left, right)            # generate_fusion.py:104: the objects' masks of the frame
        out[f"prob_{n}"] = probs.numpy().copy()
        print(f"call {n}: idx {idx} limits [{left}, {right}] -> {tuple(probs.shape)}, mean fg prob {float(probs[1:].mean()):.4f}")
    path = os.path.join(ROOT, "tests", "golden", "gen_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

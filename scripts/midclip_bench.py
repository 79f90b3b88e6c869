#!/usr/bin/env python
"""A/B of the two-stream forward / backward passes (InferenceCore.CONCURRENT_PASSES) on a mid-clip interaction of BASELINE config 3:
480x854, 70 frames, 5 objects, top_k = 50; session interact(0), interact(69), interact(35) - the third interaction propagates 34 frames
forward and 34 backward, both fused.  Prints propagated frames/s of that third interaction, sequential vs concurrent, same process,
interleaved repetitions; the masks of both orders must be identical (asserted).

    python scripts/midclip_bench.py [--reps 3] [--objects 5]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--objects", type=int, default=5)
    ap.add_argument("--frames", type=int, default=70)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    from mivos_amd.inference_core import InferenceCore
    from mivos_amd.model.fusion_net import FusionNet
    from mivos_amd.model.propagation.prop_net import PropagationNetwork
    from mivos_amd.util import synthetic
    dev = "cuda:0"
    K, T = args.objects, args.frames
    prop, fuse = PropagationNetwork(top_k=50), FusionNet()
    prop.load_state_dict(synthetic.make_prop_state(0))
    fuse.load_state_dict(synthetic.make_fuse_state(0))
    prop, fuse = prop.to(dev).eval(), fuse.to(dev).eval()
    images, gt = synthetic.synthetic_clip(T, 480, 854, K, seed=100)
    images, gt = images.to(dev), gt.to(dev)
    mid = T // 2
    res = {False: [], True: []}
    masks = {}
    for rep in range(args.reps + 1):                 # rep 0 = warm-up
        for conc in (False, True):
            core = InferenceCore(prop, fuse, images, K, mem_freq=5, device=dev)
            core.CONCURRENT_PASSES = conc
            core.interact(gt[0], 0)
            core.interact(gt[T - 1], T - 1)
            n0 = core.propagated_frames
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m = core.interact(gt[mid], mid)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if rep:
                res[conc].append((core.propagated_frames - n0) / dt)
            masks[conc] = m
        assert np.array_equal(masks[False], masks[True]), "concurrent passes changed the result"
    out = dict(workload=f"config 3 clip ({T} frames, {K} objects): third interaction at frame {mid}, {T - 3} propagated + fused frames",
               sequential_fps=[round(x, 2) for x in res[False]], concurrent_fps=[round(x, 2) for x in res[True]],
               speedup=round(float(np.median(res[True]) / np.median(res[False])), 4), identical_masks=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Conditioning probe for closed-loop parity fixtures (CPU only, oracle only - test infrastructure).

Runs the CPU oracle (== the unmodified reference, tests/test_oracle_golden.py) twice on the same clip, in fp32 and in fp64, and prints
per propagated frame how far the reference is from ITSELF: mask IoU, mismatching pixels, max |dprob|.  A fixture on which the two runs
disagree cannot decide whether a second implementation is right ("What's weak": the 70-frame headline clip).  Knobs are
post-hoc gains on the synthetic state dict (`synthetic.condition_state`) and the clip's texture; the admission bar for a fixture is
IoU(fp32, fp64) >= 0.9995 on every step.

    python scripts/studies/fixture_conditioning.py --frames 12 --height 240 --width 432 --key-gain 2 --logit-gain 3
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--objects", type=int, default=5)
    ap.add_argument("--top-k", type=int, default=50)
    ap.add_argument("--mem-freq", type=int, default=5)
    ap.add_argument("--seed", type=int, default=100)
    ap.add_argument("--key-gain", type=float, default=1.0)
    ap.add_argument("--logit-gain", type=float, default=1.0)
    ap.add_argument("--mask-gain", type=float, default=1.0)
    ap.add_argument("--logit-bias", type=float, default=0.0)
    ap.add_argument("--resid-gain", type=float, default=1.0)
    ap.add_argument("--fuse-logit-gain", type=float, default=1.0)
    ap.add_argument("--texture", type=float, default=0.0, help="amplitude of per-pixel texture added to the clip (0 = the band-limited clip)")
    ap.add_argument("--second", type=int, default=-1, help="second interaction frame (default: last; -2: none)")
    ap.add_argument("--clip-frames", type=int, default=None, help="generate a clip of this many frames and run on its first --frames (the morph / motion per step of the long clip)")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--dtypes", default="fp32,fp64")
    ap.add_argument("--kth", action="store_true", help="print the softmax weight of the k-th survivor (quantiles over queries) per frame")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    torch.set_num_threads(args.threads)
    from oracle import stm_oracle as O
    from mivos_amd.util import synthetic
    from mivos_amd.util.tensor_util import compute_np_iou
    images, gt = synthetic.synthetic_clip(args.clip_frames or args.frames, args.height, args.width, args.objects, seed=args.seed, texture=args.texture)
    images, gt = images[:, :args.frames], gt[:args.frames]
    sd = synthetic.condition_state(synthetic.make_prop_state(0), key_gain=args.key_gain, logit_gain=args.logit_gain, mask_gain=args.mask_gain, logit_bias=args.logit_bias, resid_gain=args.resid_gain)
    fsd = synthetic.condition_fuse_state(synthetic.make_fuse_state(0), logit_gain=args.fuse_logit_gain)
    inter = [0] + ([] if args.second == -2 else [args.frames - 1 if args.second == -1 else args.second])
    runs = {}
    kth = {}
    if args.kth:
        orig = O.topk_softmax

        def spy(aff, top_k):
            a, values, indices = orig(aff, top_k)
            e = torch.exp(values - values[:, 0])
            w = (e / e.sum(dim=1, keepdim=True))[0, -1]          # weight of the k-th survivor per query
            kth.setdefault("w", []).append(w.float())
            return a, values, indices
        O.topk_softmax = spy
    for name in args.dtypes.split(","):
        dt = torch.float64 if name == "fp64" else torch.float32
        core = O.OracleCore(sd, fsd, images, args.objects, mem_freq=args.mem_freq, top_k=args.top_k, dtype=dt)
        t0 = time.perf_counter()
        res = []
        for idx in inter:
            m = core.interact(gt[idx], idx).copy()
            res.append((m, core.prob.clone()))
            print(f"{name}: interact({idx}) done, {core.propagated} frames, {time.perf_counter() - t0:.0f} s", flush=True)
        runs[name] = res
        if args.kth and kth:
            w = torch.cat(kth.pop("w"))
            qs = torch.quantile(w[torch.randperm(w.numel())[:200000]], torch.tensor([0.5, 0.9, 0.99, 1.0]))
            print(f"{name}: k-th survivor softmax weight, median / 90 % / 99 % / max over queries: " + " / ".join(f"{float(q):.2e}" for q in qs))
    if len(runs) < 2:
        return
    a, b = [runs[n] for n in args.dtypes.split(",")[:2]]
    K = args.objects
    out = []
    for n, idx in enumerate(inter):
        (ma, pa), (mb, pb) = a[n], b[n]
        worst = 1.0
        for t in range(args.frames):
            iou = float(np.mean([compute_np_iou(ma[t] == j, mb[t] == j) for j in range(1, K + 1)]))
            area = [int((mb[t] == j).sum()) for j in range(1, K + 1)]
            d = float((pa[:, t].double() - pb[:, t].double()).abs().max())
            mism = int((ma[t] != mb[t]).sum())
            out.append(dict(interaction=idx, frame=t, iou=iou, mismatch=mism, dprob=d, area=area))
            worst = min(worst, iou)
            print(f"interact({idx}) frame {t:3d}  IoU {iou:.6f}  mismatch {mism:5d} px  max|dprob| {d:.2e}  areas {area}", flush=True)
        print(f"interact({idx}): min IoU {worst:.6f}")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(args=vars(args), frames=out), f)


if __name__ == "__main__":
    main()

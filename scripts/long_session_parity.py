#!/usr/bin/env python
"""Long closed-loop parity run: BASELINE config 3's FULL session (480x854, 70 frames, 5 objects, top_k = 50, mem_freq = 5,
interact(0) = 69 propagated frames, then interact(69) = 68 propagated + fused frames) on the engine against the CPU oracle in
fp32 (== the unmodified reference bit for bit, tests/test_oracle_golden.py) and, optionally, in fp64 (arbitration).  Per frame
and interaction: mask IoU, mismatching pixels, max / quantiles of |dprob|.  Too long for `pytest -m gpu` (137 CPU frames of ~3 s
in fp32, ~3x that in fp64); run once per round under gpurun and commit the JSON under profiles/.

    python scripts/long_session_parity.py oracle --dtype fp32 --out /tmp/long32     # CPU only (can run beside GPU work)
    python scripts/long_session_parity.py oracle --dtype fp64 --out /tmp/long64
    python scripts/long_session_parity.py engine --ref32 /tmp/long32 [--ref64 /tmp/long64] --json gpurun_out/long_session_parity.json

    python scripts/long_session_parity.py pack --out gpurun_in/long32        # shrink an oracle directory for shipping to the GPU box

The oracle phase writes prob_<n>.npy / masks_<n>.npy after every interaction and a `done` marker.  The CPU oracles are slow (fp32 ~12 min,
fp64 ~35 min on 8 cores) and need no GPU, so they run on the builder's machine and travel with the repo snapshot (gpurun_in/, git-ignored);
`pack` keeps the masks at full resolution (IoU is exact) and every SUB-th pixel in both directions of the probabilities as float32 (|dprob|
maxima and quantiles are then statistics over 1/SUB^2 of the pixels: 26 k of 415 k per frame and channel at SUB = 4; the full tensors are
0.7 GB per interaction and precision)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(frames=70, height=480, width=854, objects=5, top_k=50, mem_freq=5, seed=100, interactions=[0, 69],
           # conditioning of the fixture (synthetic.condition_state / synthetic_clip(texture=...)).  Round 5: logit gain 0.6, mask gain 0.2 - admitted because
           # the reference's own fp32 and fp64 runs agree to IoU >= 0.9995 at every one of the 137 steps on it (all gains 1 = the round-4 fixture: 0.927)
           key_gain=1.0, logit_gain=0.6, mask_gain=0.2, logit_bias=0.0, fuse_logit_gain=1.0, texture=0.0, clip_frames=None)


def clip(cfg):
    """The clip: `frames` frames; with clip_frames = N > frames the first `frames` of an N-frame clip (the per-step motion of the long clip)."""
    from mivos_amd.util import synthetic
    images, gt = synthetic.synthetic_clip(cfg["clip_frames"] or cfg["frames"], cfg["height"], cfg["width"], cfg["objects"], seed=cfg["seed"], texture=cfg["texture"])
    return images[:, :cfg["frames"]], gt[:cfg["frames"]]


def states(cfg):
    from mivos_amd.util import synthetic
    return (synthetic.condition_state(synthetic.make_prop_state(0), key_gain=cfg["key_gain"], logit_gain=cfg["logit_gain"], mask_gain=cfg["mask_gain"], logit_bias=cfg["logit_bias"]),
            synthetic.condition_fuse_state(synthetic.make_fuse_state(0), logit_gain=cfg["fuse_logit_gain"]))


def run_oracle(args, cfg):
    from oracle import stm_oracle as O
    from mivos_amd.util import synthetic
    torch.set_num_threads(args.threads)
    os.makedirs(args.out, exist_ok=True)
    dt = torch.float64 if args.dtype == "fp64" else torch.float32
    images, gt = clip(cfg)
    sd, fsd = states(cfg)
    core = O.OracleCore(sd, fsd, images, cfg["objects"], mem_freq=cfg["mem_freq"], top_k=cfg["top_k"], dtype=dt, record_margins=True)
    t0 = time.perf_counter()
    for n, idx in enumerate(cfg["interactions"]):
        masks = core.interact(gt[idx], idx)
        np.save(os.path.join(args.out, f"masks_{n}.npy"), masks)
        np.save(os.path.join(args.out, f"prob_{n}.npy"), core.prob.numpy())
        with open(os.path.join(args.out, f"margins_{n}.json"), "w") as f:
            json.dump({str(k): v for k, v in core.topk_margin.items()}, f)
        print(f"oracle {args.dtype}: interact({idx}) done, {core.propagated} frames, {time.perf_counter() - t0:.0f} s", flush=True)
    with open(os.path.join(args.out, "done"), "w") as f:
        json.dump(dict(seconds=time.perf_counter() - t0, frames=core.propagated, threads=args.threads, dtype=args.dtype), f)


SUB = 4


def run_pack(args, cfg):
    d = args.out
    n = 0
    while os.path.exists(os.path.join(d, f"prob_{n}.npy")):
        p = np.load(os.path.join(d, f"prob_{n}.npy"), mmap_mode="r")
        np.save(os.path.join(d, f"probsub_{n}.npy"), np.ascontiguousarray(p[..., ::SUB, ::SUB]).astype(np.float32))
        np.savez_compressed(os.path.join(d, f"masksz_{n}.npz"), m=np.load(os.path.join(d, f"masks_{n}.npy")))
        os.remove(os.path.join(d, f"prob_{n}.npy"))
        os.remove(os.path.join(d, f"masks_{n}.npy"))
        n += 1
    print("packed", n, "interactions in", d, sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) // (1 << 20), "MiB")


def load_ref(d, n):
    """(masks uint8 [T,H,W], prob float32 tensor, subsampled?) of interaction n from a full or a packed oracle directory."""
    if os.path.exists(os.path.join(d, f"probsub_{n}.npy")):
        return np.load(os.path.join(d, f"masksz_{n}.npz"))["m"], torch.from_numpy(np.load(os.path.join(d, f"probsub_{n}.npy"))), True
    return np.load(os.path.join(d, f"masks_{n}.npy")), torch.from_numpy(np.load(os.path.join(d, f"prob_{n}.npy"))), False


def quantiles(x, qs=(0.999, 0.9999)):
    flat = x.flatten()
    return [float(flat.kthvalue(max(1, int(round(flat.numel() * q)))).values) for q in qs]


def run_engine(args, cfg):
    from mivos_amd.inference_core import InferenceCore
    from mivos_amd.model.fusion_net import FusionNet
    from mivos_amd.model.propagation.prop_net import PropagationNetwork
    from mivos_amd.util import synthetic
    from mivos_amd.util.tensor_util import compute_np_iou
    torch.set_grad_enabled(False)
    dev = "cuda:0"
    from mivos_amd import ops
    ops.CONV_PRECISION = args.precision
    ops.AFFINITY_PRECISION = args.affinity
    if args.no_act_path:
        ops.USE_ACT_PATH = False
    if args.no_stem_kernel:
        from mivos_amd.model.propagation import modules
        modules.STEM_FROM_PLANES = False
    if args.no_cout1_projection:
        ops.COUT1_PROJECTION = False
    t_wait = time.time()
    for d in (args.ref32, args.ref64):
        while d and not os.path.exists(os.path.join(d, "done")):
            if time.time() - t_wait > args.wait:
                if d == args.ref64:
                    print(f"fp64 oracle not finished after {args.wait} s: continuing without arbitration", flush=True)
                    args.ref64 = None
                    break
                raise SystemExit(f"oracle results missing in {d}")
            time.sleep(5)
    K = cfg["objects"]
    prop, fuse = PropagationNetwork(top_k=cfg["top_k"]), FusionNet()
    sd, fsd = states(cfg)
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    prop, fuse = prop.to(dev).eval(), fuse.to(dev).eval()
    images, gt = clip(cfg)
    core = InferenceCore(prop, fuse, images, K, mem_freq=cfg["mem_freq"], device=dev)
    out = dict(config=cfg, engine_precision=args.precision, affinity_precision=ops.affinity_precision(), act_path=ops.act_path(), toggles=dict(no_stem_kernel=args.no_stem_kernel, no_cout1_projection=args.no_cout1_projection), oracle_fp32=json.load(open(os.path.join(args.ref32, "done"))),
               oracle_fp64=json.load(open(os.path.join(args.ref64, "done"))) if args.ref64 else None, interactions=[])
    for n, idx in enumerate(cfg["interactions"]):
        t0 = time.perf_counter()
        masks = core.interact(gt[idx], idx)
        torch.cuda.synchronize()
        secs = time.perf_counter() - t0
        ref_m, ref_p, sub = load_ref(args.ref32, n)
        eng_p = core.prob.cpu()
        if sub:
            eng_p = eng_p[..., ::SUB, ::SUB].contiguous()
        m64, p64, sub64 = load_ref(args.ref64, n) if args.ref64 else (None, None, sub)
        assert sub64 == sub
        out["probability_pixels_compared"] = "every %d-th pixel in both directions" % SUB if sub else "all"
        margins = json.load(open(os.path.join(args.ref32, f"margins_{n}.json")))
        frames = []
        for t in range(cfg["frames"]):
            iou = float(np.mean([compute_np_iou(masks[t] == j, ref_m[t] == j) for j in range(1, K + 1)]))
            d = (eng_p[:, t] - ref_p[:, t]).abs()
            q = quantiles(d)
            rec = dict(frame=t, iou=round(iou, 6), mismatch_px=int((masks[t] != ref_m[t]).sum()), dprob_max=float(d.max()), dprob_q999=q[0], dprob_q9999=q[1],
                       frac_gt_1e3=float((d > 1e-3).float().mean()), topk_margin_fp32=margins.get(str(t)))
            if p64 is not None:
                e = (eng_p[:, t].double() - p64[:, t]).abs()
                r = (ref_p[:, t].double() - p64[:, t]).abs()
                rec.update(engine_vs_fp64_max=float(e.max()), ref32_vs_fp64_max=float(r.max()), engine_vs_fp64_q999=quantiles(e)[0], ref32_vs_fp64_q999=quantiles(r)[0],
                           iou_engine_vs_fp64=round(float(np.mean([compute_np_iou(masks[t] == j, m64[t] == j) for j in range(1, K + 1)])), 6),
                           iou_ref32_vs_fp64=round(float(np.mean([compute_np_iou(ref_m[t] == j, m64[t] == j) for j in range(1, K + 1)])), 6))
            frames.append(rec)
        live = [f for f in frames if f["frame"] not in cfg["interactions"][:n + 1]]
        far = frames[cfg["frames"] - 1] if idx == 0 else frames[1]
        summ = dict(interact=idx, engine_seconds=round(secs, 3), min_iou=min(f["iou"] for f in live), mean_iou=round(float(np.mean([f["iou"] for f in live])), 6),
                    iou_at_farthest_frame=far["iou"], farthest_frame=far["frame"], frames_below_0999=[f["frame"] for f in live if f["iou"] < 0.999],
                    total_mismatch_px=sum(f["mismatch_px"] for f in live), pixels_per_frame=cfg["height"] * cfg["width"],
                    max_dprob=max(f["dprob_max"] for f in live), worst_q999=max(f["dprob_q999"] for f in live), worst_q9999=max(f["dprob_q9999"] for f in live))
        if p64 is not None:
            summ.update(min_iou_engine_vs_fp64=min(f["iou_engine_vs_fp64"] for f in live), min_iou_ref32_vs_fp64=min(f["iou_ref32_vs_fp64"] for f in live),
                        max_engine_vs_fp64=max(f["engine_vs_fp64_max"] for f in live), max_ref32_vs_fp64=max(f["ref32_vs_fp64_max"] for f in live),
                        worst_q999_engine_vs_fp64=max(f["engine_vs_fp64_q999"] for f in live), worst_q999_ref32_vs_fp64=max(f["ref32_vs_fp64_q999"] for f in live),
                        # e / r per frame: how far the engine is from the fp64 run relative to how far the reference's own fp32 run is
                        median_ratio_of_maxima=round(float(np.median([f["engine_vs_fp64_max"] / max(f["ref32_vs_fp64_max"], 1e-12) for f in live])), 3),
                        median_ratio_of_q999=round(float(np.median([f["engine_vs_fp64_q999"] / max(f["ref32_vs_fp64_q999"], 1e-12) for f in live])), 3),
                        worst_ratio_of_maxima=round(float(max(f["engine_vs_fp64_max"] / max(f["ref32_vs_fp64_max"], 1e-12) for f in live)), 3),
                        frames_ref32_vs_fp64_below_09995=[f["frame"] for f in live if f["iou_ref32_vs_fp64"] < 0.9995])
        print(json.dumps(summ), flush=True)
        out["interactions"].append(dict(summary=summ, frames=frames))
    os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
    with open(args.json, "w") as f:
        json.dump(out, f)
    print("wrote", args.json)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("phase", choices=("oracle", "engine", "pack"))
    ap.add_argument("--dtype", default="fp32", choices=("fp32", "fp64"))
    ap.add_argument("--out", default="/tmp/long32")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--ref32", default="/tmp/long32")
    ap.add_argument("--ref64", default=None)
    ap.add_argument("--wait", type=int, default=1800, help="engine phase: seconds to wait for the oracle results")
    ap.add_argument("--json", default=os.path.join(ROOT, "gpurun_out", "long_session_parity.json"))
    ap.add_argument("--precision", default="f16x3", choices=("f16x3", "f32"),
                    help="engine phase: ops.CONV_PRECISION - f16x3 (the default engine) or f32 (every convolution and the affinity on exact fp32 MFMA)")
    ap.add_argument("--affinity", default=None, choices=("f16x3", "f32"), help="engine phase: pin the memory-read affinity's arithmetic (default: follows --precision)")
    ap.add_argument("--no-act-path", action="store_true", help="engine phase (diagnostics): every convolution on the register-staged f16x3 kernels from fp32 activations (no SH32 tensors)")
    ap.add_argument("--no-stem-kernel", action="store_true", help="engine phase (diagnostics): the 7x7 stems on the generic convolution kernels")
    ap.add_argument("--no-cout1-projection", action="store_true", help="engine phase (diagnostics): decoder.pred on the generic kernels instead of projection + tap sum")
    for k in ("key_gain", "logit_gain", "mask_gain", "logit_bias", "fuse_logit_gain", "texture"):
        ap.add_argument("--" + k.replace("_", "-"), type=float, default=None)
    ap.add_argument("--clip-frames", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None, help="shorter clip (smoke runs)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--objects", type=int, default=None)
    args = ap.parse_args()
    cfg = dict(CFG)
    for k in ("frames", "height", "width", "objects", "key_gain", "logit_gain", "mask_gain", "logit_bias", "fuse_logit_gain", "texture", "clip_frames"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    cfg["interactions"] = [0, cfg["frames"] - 1]
    torch.set_grad_enabled(False)
    dict(oracle=run_oracle, engine=run_engine, pack=run_pack)[args.phase](args, cfg)


if __name__ == "__main__":
    main()

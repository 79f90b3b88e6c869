#!/usr/bin/env python
"""Benchmark of the MI355X propagation + fusion engine (BASELINE.json metric: propagated frames/sec,
DAVIS-2017 480p multi-object).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--objects 5] [--height 480 --width 854]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one propagated frame = one iteration of the reference's do_pass loop
(`inference_core.py:165`): query features (cached per frame), per-object top-k memory read, mask decoder,
aggregation, memorize into the bank, and — on the second interaction — difference-aware fusion.
Workload (BASELINE config 3): one synthetic 480x854 clip per GPU, K objects, top_k=50, mem_freq=5;
`interact(mask, 0)` propagates T-1 frames, `interact(mask, T-1)` re-propagates T-2 frames with fusion.
T is sized so that the session has >= warmup + steps propagated frames; the timed region is exactly
`steps` of them, bracketed by barrier + torch.cuda.synchronize().  All inputs are resident in HBM
before the timed region (mem_profile 0).  Multi-GPU: sequences shard (one clip per rank, no data-path
collective), value = frames of all ranks / max-over-ranks time, scaling = weak.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
# fp16 MFMA dense peak (2.5 PFLOP/s) / 3 MFMA products per algorithmic multiply-add of the error-compensated
# f16x3 convolution = the roofline of that kernel in ALGORITHMIC (fp32-equivalent) FLOP/s
F16X3_PEAK_TFLOPS = 2500.0 / 3
VARIANT_NAMES = {0: "conv_igemm_kernel<128,128,2,2>", 1: "conv_igemm_kernel<64,64,2,2>",
                 2: "conv_igemm_kernel<128,32,4,1>", 3: "conv_igemm_kernel<128,64,2,2>", 4: "conv_cout1_kernel",
                 10: "conv_f16x3_kernel<128,128,2,2>", 11: "conv_f16x3_kernel<64,64,2,2>", 12: "conv_f16x3_kernel<128,32,4,1>",
                 13: "conv_f16x3_kernel<128,64,2,2>", 14: "conv_cout1_kernel", 15: "conv_f16x3_pipe_kernel<256,256,2,4>",
                 16: "conv_f16x3_pipe_kernel<128,256,2,4>", 17: "conv_f16x3_pipe_kernel<128,128,4,2>",
                 18: "conv_f16x3_pipe_kernel<64,256,2,4>", 19: "conv3x3_n32_direct_kernel",
                 20: "conv_f16x3_pp_kernel<128,128,2,4,0>", 21: "conv_f16x3_pp_kernel<128,256,2,4,0>",
                 22: "conv_f16x3_pp_kernel<128,64,4,2,0>", 23: "conv_f16x3_pp_kernel<256,256,2,4,0>"}


class StepTimer:
    """step_cb hook: synchronises + stamps the clock exactly at step `warmup` and `warmup + steps`."""

    def __init__(self, warmup, steps, profile_every, ops, shard):
        self.warmup, self.steps, self.every, self.ops, self.shard = warmup, steps, profile_every, ops, shard
        self.n, self.t0, self.t1, self.samples = 0, None, None, []
        self._arm()

    def _arm(self):
        if self.n == self.warmup and self.t0 is None:
            torch.cuda.synchronize()
            self.shard.barrier()
            torch.cuda.synchronize()
            self.t0 = time.perf_counter()
        timed = self.t0 is not None and self.t1 is None
        # sample every `every`-th timed step with HIP events around each conv launch
        self.ops.PROFILE = self.samples if (timed and self.every and (self.n - self.warmup) % self.every == 0) else None

    def __call__(self):
        self.n += 1
        if self.n == self.warmup + self.steps and self.t1 is None:
            torch.cuda.synchronize()
            self.shard.barrier()
            torch.cuda.synchronize()
            self.t1 = time.perf_counter()
        self._arm()


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (profiles/*pmc_traffic.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench command, FETCH_SIZE x2 per the gfx950
    correction of MI355X_MICROARCH.md §HBM).  None if no committed measurement matches."""
    import glob
    import re
    key = re.sub(r"[ ,]", "", kernel_name)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            table = json.load(open(path))
        except Exception:
            continue
        for name, rec in table.items():
            norm = re.sub(r"[ ,]", "", name.replace("mivos::", ""))
            if norm.startswith(key.rstrip(">")):
                return dict(bytes_per_launch=int(rec["read_bytes_per_launch"] + rec["write_bytes_per_launch"]),
                            read=int(rec["read_bytes_per_launch"]), write=int(rec["write_bytes_per_launch"]),
                            source=os.path.basename(path))
    return None


def event_pair_overhead():
    """Seconds a HIP-event pair measures around NOTHING on a busy stream (timestamp writes + command-processor gaps):
    subtracted from every per-launch sample so that short launches (50 us) are not inflated by 10-15 %.  Calibrated with
    a kernel in front of every pair, as in the sampled steps."""
    x = torch.zeros(1 << 20, device="cuda")
    pairs = []
    for _ in range(64):
        x.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in pairs)
    return t[len(t) // 2] * 1e-3


def conv_roofline(samples, overhead=0.0):
    """Aggregate the HIP-event samples per kernel instantiation; the dominant one is the roofline kernel."""
    agg = {}
    for variant, flops, e0, e1, shape in samples:
        a = agg.setdefault(variant, [0.0, 0.0, 0, 0.0])
        a[0] += flops
        a[1] += max(e0.elapsed_time(e1) * 1e-3 - overhead, 1e-7)
        a[2] += 1
        m, cin, cout, k, stride, has_res = shape
        # one read of the fp32 input, the (hi, lo) fp16 weights and the residual, one write of the fp32 output
        a[3] += 4.0 * m * stride * stride * cin + 4.0 * cout * k * k * cin + 4.0 * m * cout * (2 if has_res else 1)
    if not agg:
        return None, {}
    table = {VARIANT_NAMES[v]: dict(launches=a[2], avg_us=round(a[1] / a[2] * 1e6, 2), tflops=round(a[0] / a[1] / 1e12, 2),
                                    time_share=round(a[1] / sum(x[1] for x in agg.values()), 3)) for v, a in sorted(agg.items())}
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    v, (flops, secs, n, abytes) = dom
    ach = flops / secs / 1e12
    peak = F16X3_PEAK_TFLOPS if v >= 10 else MFMA_F32_PEAK_TFLOPS
    roof = dict(bound="mfma", kernel=VARIANT_NAMES[v], achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s",
                frac=round(ach / peak, 4), traffic=pmc_traffic(VARIANT_NAMES[v]), launches_sampled=n,
                peak_note=("algorithmic (fp32-equivalent) FLOP/s; kernel issues 3 fp16 MFMA products per term: 2500/3"
                           if v >= 10 else "fp32 MFMA dense peak"),
                avg_launch_us=round(secs / n * 1e6, 2), algorithmic_gflop_per_launch=round(flops / n / 1e9, 3),
                algorithmic_bytes_per_launch=int(abytes / n))
    return roof, table


def cpu_baseline(images, gt, k, top_k, mem_freq, engine_masks, frames, with_fp64=True):
    """The CPU oracle (restatement of the reference, oracle/stm_oracle.py) on a bounded sample of the same
    workload: the first `frames` propagated frames of the first interaction.  Parity is reported three ways:
    engine vs the fp32 oracle, and - because the algorithm is closed-loop and discontinuous (argmax / top-k
    on an untrained network) - both of them against an fp64 run of the same oracle, which is the
    noise floor any fp32 implementation of the reference has on this clip."""
    from oracle import stm_oracle as O
    from mivos_amd.util import synthetic
    from mivos_amd.util.tensor_util import compute_np_iou
    # oneDNN/OpenMP scale poorly past a few dozen threads on these small convolutions (256 threads on the
    # 256-core host of the GPU box were 10x slower than 8 threads): use min(32, cores) and say so
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd, fsd = synthetic.make_prop_state(0), synthetic.make_fuse_state(0)
    core = O.OracleCore(sd, fsd, images[:, :frames + 1], k, mem_freq=mem_freq, top_k=top_k)
    t0 = time.perf_counter()
    ref = core.interact(gt[0], 0)
    dt = time.perf_counter() - t0

    def miou(a, b):
        return round(float(sum(compute_np_iou(a == j, b == j) for j in range(1, k + 1)) / k), 6)

    eng = engine_masks[:frames + 1]
    parity = dict(frames=frames, mean_iou_engine_vs_ref_fp32=miou(eng[1:], ref[1:]),
                  mismatching_pixel_fraction=round(float((eng[1:] != ref[1:]).mean()), 6))
    if with_fp64:
        c64 = O.OracleCore(sd, fsd, images[:, :frames + 1], k, mem_freq=mem_freq, top_k=top_k, dtype=torch.float64)
        r64 = c64.interact(gt[0], 0)
        parity.update(mean_iou_ref_fp32_vs_ref_fp64=miou(ref[1:], r64[1:]), mean_iou_engine_vs_ref_fp64=miou(eng[1:], r64[1:]))
    return dict(value=round(frames / dt, 4), unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"first {frames} propagated frames of the same clip ({k} objects, no fusion), oracle/stm_oracle.py on PyTorch-CPU fp32",
                seconds=round(dt, 2)), parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--objects", type=int, default=5)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--top-k", type=int, default=50)
    ap.add_argument("--mem-freq", type=int, default=5)
    ap.add_argument("--cpu-frames", type=int, default=2, help="propagated frames of the CPU-oracle sample (0 = skip)")
    ap.add_argument("--profile-every", type=int, default=7,
                    help="HIP-event sample every n-th timed step (0 = off); co-prime with InferenceCore.QUERY_BATCH "
                         "so the samples see every phase of the batched query encoding")
    args = ap.parse_args()

    torch.set_grad_enabled(False)
    from mivos_amd import ops, shard
    from mivos_amd.inference_core import InferenceCore
    from mivos_amd.model.fusion_net import FusionNet
    from mivos_amd.model.propagation.prop_net import PropagationNetwork
    from mivos_amd.util import synthetic

    rank, world, local = shard.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"

    K, need = args.objects, args.warmup + args.steps
    T = max(4, (need + 3 + 1) // 2)                       # session has 2T-3 propagated frames
    prop, fuse = PropagationNetwork(top_k=args.top_k), FusionNet()
    prop.load_state_dict(synthetic.make_prop_state(0))
    fuse.load_state_dict(synthetic.make_fuse_state(0))
    images, gt = synthetic.synthetic_clip(T, args.height, args.width, K, seed=100 + rank)
    core = InferenceCore(prop.eval(), fuse.eval(), images, K, mem_profile=0, mem_freq=args.mem_freq, device=dev)

    timer = StepTimer(args.warmup, args.steps, args.profile_every, ops, shard)
    masks_first = core.interact(gt[0], 0, step_cb=timer).copy()
    core.interact(gt[T - 1], T - 1, step_cb=timer)
    ops.PROFILE = None
    torch.cuda.synchronize()
    assert timer.t0 is not None and timer.t1 is not None and core.propagated_frames >= need, (core.propagated_frames, need)
    elapsed = shard.max_over_ranks(timer.t1 - timer.t0, device=dev)
    recs = shard.gather_records([dict(rank=rank, frames=args.steps, seconds=timer.t1 - timer.t0)])

    if rank != 0:
        return
    ev_overhead = event_pair_overhead()
    roof, table = conv_roofline(timer.samples, ev_overhead)
    if roof is not None:
        roof["event_pair_overhead_us"] = round(ev_overhead * 1e6, 2)
    if os.environ.get("MIVOS_BENCH_SHAPES"):          # debug: per-shape conv time inside the timed region
        agg = {}
        for variant, flops, e0, e1, shape in timer.samples:
            a = agg.setdefault((variant,) + shape, [0.0, 0.0, 0])
            a[0] += flops; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        tot = sum(a[1] for a in agg.values())
        for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f"# {VARIANT_NAMES[key[0]]:38s} M={key[1]:7d} Cin={key[2]:4d} Cout={key[3]:4d} k={key[4]} s={key[5]}  n={a[2]:4d} "
                  f"avg {a[1] / a[2] * 1e6:8.1f} us  {a[0] / a[1] / 1e12:6.1f} TF/s  share {a[1] / tot * 100:5.1f}%", file=sys.stderr)
    out = dict(metric="propagated frames/sec, DAVIS-2017 480p multi-object", value=round(world * args.steps / elapsed, 3),
               unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=f"davis480p_multiobject_fusion: {args.height}x{args.width} clip of {T} frames per GPU, "
                                    f"{K} objects, top_k={args.top_k}, mem_freq={args.mem_freq}, interact(0) then interact({T - 1}) "
                                    f"(fused re-propagation); timed steps {args.warmup}..{args.warmup + args.steps} of {2 * T - 3}",
                           objects=K, frames=T, height=args.height, width=args.width, top_k=args.top_k, mem_freq=args.mem_freq,
                           plain_frames_timed=max(0, min(T - 1, need) - args.warmup), parallelism=f"sequence-sharded x{world}"),
               roofline=roof, conv_kernels=table, per_rank=recs)
    if world == 1 and args.cpu_frames > 0:
        out["cpu_baseline"], out["parity"] = cpu_baseline(images, gt, K, args.top_k, args.mem_freq, masks_first, args.cpu_frames)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Suite runner: many independent clips, sharded over the GPUs of a node.

The reference evaluates a dataset sequence by sequence with a fresh processor per sequence
(`eval_interactive_davis.py:74-108`; the YouTube-VOS loader resizes every clip so that its short side is 480,
`dataset/yv_test_dataset.py:102-109`).  Sequences never exchange data, so the multi-GPU form is: one process per GPU,
weights replicated, clips assigned longest-first (``shard.assign_sequences``), NO data-path collective, one small
``all_gather_object`` of per-clip records at the end.  This module is that loop for the MI355X engine:

    specs   = synthetic_suite(474)                       # or any list of ClipSpec
    records = run_suite(specs, engine_factory, rank, world)
    summary = summarize(gather_records(records))

``engine_factory(spec) -> (core, first_mask)`` builds the per-clip engine (an ``InferenceCore`` over the clip's
frames) — injected so that the sharding logic is testable on CPU with a stub engine (tests/test_eval_suite_gloo.py).
"""
import time
import zlib
from collections import namedtuple

import numpy as np

from . import shard

ClipSpec = namedtuple("ClipSpec", "clip_id frames objects height width seed")


def yv_480p_size(h, w):
    """Target size of the YouTube-VOS loader (yv_test_dataset.py:102-109): short side -> 480, the other side scaled
    with integer floor division."""
    return (h * 480 // w, 480) if h > w else (480, w * 480 // h)


def synthetic_suite(n_clips=474, seed=0, source_sizes=((720, 1280),)):
    """BASELINE config 4 restated as synthetic input: 474 clips (the size of YouTube-VOS-2018 val),
    lengths 5 * U{4..36} frames, K ~ U{1..5} objects, source frames resized by the loader's rule (720x1280 -> 480x853)."""
    r = np.random.RandomState(0xC0FFEE ^ seed)
    specs = []
    for i in range(n_clips):
        h, w = source_sizes[int(r.randint(len(source_sizes)))]
        th, tw = yv_480p_size(h, w)
        specs.append(ClipSpec(i, int(5 * r.randint(4, 37)), int(r.randint(1, 6)), th, tw, 1000 + i))
    return specs


def mask_checksum(masks):
    """Order-independent identity of a clip's output (uint8 [T,H,W]) for cross-run / cross-world-size comparisons."""
    return int(zlib.crc32(np.ascontiguousarray(masks).tobytes()))


def run_suite(specs, engine_factory, rank=0, world=1, sync=None, on_clip=None, lanes=1, lane_ctx=None):
    """Process this rank's share of `specs`.  Returns one record per processed clip:
    dict(clip, rank, frames (propagated), objects, seconds, checksum).  `sync()` (e.g. torch.cuda.synchronize) brackets
    the per-clip timer; `on_clip(spec, masks)` receives every result (mask egress).

    lanes > 1: that many clips of this rank are in flight at once, advanced in turn frame by frame (`core.interact_steps`),
    lane i inside the context `lane_ctx(i)` - a HIP stream per lane (`stream_lanes`), so that the under-filled launches of one
    clip (a 480p frame of 1-5 objects fills a fraction of the 256 CUs at the 1/16-resolution layers) run beside the other's.
    Clips stay independent - no tensor is shared between lanes but the read-only weights - and the results are bit-identical
    to a lanes = 1 run (tests; since round 6 the launch-geometry hint ops.chip_share no longer enters the split-K slicing).  A clip's `seconds` is then its share of the rank's wall clock (its own in-flight time, scaled so
    that the clips of a rank add up to the rank's wall clock), which keeps `summarize` and the cost-model fit meaningful."""
    parts = shard.assign_sequences([shard.clip_cost(s.frames - 1, s.objects) for s in specs], world)
    if lanes <= 1:
        records = []
        for i in parts[rank]:
            spec = specs[i]
            core, first_mask = engine_factory(spec)
            if sync:
                sync()
            t0 = time.perf_counter()
            masks = core.interact(first_mask, 0)
            if sync:
                sync()
            dt = time.perf_counter() - t0
            if on_clip is not None:
                on_clip(spec, masks)
            records.append(dict(clip=spec.clip_id, rank=rank, frames=spec.frames - 1, objects=spec.objects,
                                seconds=dt, checksum=mask_checksum(masks)))
            del core
        return records
    import contextlib
    if lane_ctx is None:
        lane_ctx = lambda lane: contextlib.nullcontext()
    todo = list(parts[rank])
    records, active = {}, {}                    # lane -> [spec, core, generator, t0]
    if sync:
        sync()
    t_start = time.perf_counter()
    share = getattr(lane_ctx, "chip_share", None)           # stream_lanes: tells the convolutions how many streams share the GPU
    with (share(lanes) if share else contextlib.nullcontext()):
        _advance_lanes(specs, engine_factory, rank, lanes, lane_ctx, on_clip, todo, active, records)
    if sync:
        sync()
    wall = time.perf_counter() - t_start
    out = [records[specs[i].clip_id] for i in parts[rank]]
    in_flight = sum(r["seconds"] for r in out)
    for r in out:
        r["seconds"] = r["seconds"] * wall / in_flight if in_flight > 0 else 0.0
    return out


def _advance_lanes(specs, engine_factory, rank, lanes, lane_ctx, on_clip, todo, active, records):
    while todo or active:
        for lane in range(lanes):
            with lane_ctx(lane):
                if lane not in active:
                    if not todo:
                        continue
                    spec = specs[todo.pop(0)]
                    core, first_mask = engine_factory(spec)
                    active[lane] = [spec, core, core.interact_steps(first_mask, 0), time.perf_counter()]
                spec, core, gen, t0 = active[lane]
                try:
                    next(gen)
                except StopIteration as done:            # the clip's masks are on the host (interact's final copy waits for this lane's stream only)
                    masks = done.value
                    if on_clip is not None:
                        on_clip(spec, masks)
                    records[spec.clip_id] = dict(clip=spec.clip_id, rank=rank, frames=spec.frames - 1, objects=spec.objects,
                                                 seconds=time.perf_counter() - t0, checksum=mask_checksum(masks), lanes=lanes)
                    del active[lane]


_LANE_STREAMS = {}


def stream_lanes(device, lanes):
    """`lane_ctx` of run_suite for a GPU: one HIP stream per lane (created once), entered with torch.cuda.stream; its `chip_share`
    attribute (ops.chip_share) tells every convolution launched inside how many streams share the chip."""
    import torch
    from . import ops
    key = (str(device), lanes)
    if key not in _LANE_STREAMS:                # once per process: the allocator pools and per-stream workspaces stay warm from suite to suite
        _LANE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(lanes)]
    streams = _LANE_STREAMS[key]

    def ctx(lane):
        return torch.cuda.stream(streams[lane])
    ctx.chip_share = ops.chip_share
    return ctx


# ---- the same loop over a real dataset (mivos_amd/dataset/*: the reference's test-time loaders) -----------------------------------

def dataset_suite(dataset):
    """One ClipSpec per video of a `DAVISTestDataset` / `YouTubeVOSTestDataset` from its metadata only (no frame is decoded): clip_id = the
    dataset index, frames, objects (DAVIS: from the first annotation; YouTube-VOS: the labels of its first annotated frame - later objects
    only shift the cost estimate), frame size after the loader's resize rule.  Feeds `run_suite(specs, dataset_factory(...), rank, world)`."""
    specs = []
    for i, name in enumerate(dataset.videos):
        if hasattr(dataset, "num_frames"):                                   # DAVIS
            h, w = dataset.shape[name][-2:]
            specs.append(ClipSpec(i, int(dataset.num_frames[name]), max(1, int(dataset.num_objects[name])), int(h), int(w), -1))
        else:                                                                # YouTube-VOS: short side -> 480 (yv_test_dataset.py:102-109)
            h, w = dataset.shape[name][-2:]
            th, tw = yv_480p_size(int(h), int(w))
            specs.append(ClipSpec(i, len(dataset.frames[name]), 1, th, tw, -1))
    return specs


def first_frame_mask(gt):
    """gt [K,T,1,H,W] one-hot objects of a loader's dictionary -> the `interact` argument of frame 0: [K+1,1,H,W], background first
    (every pixel no object claims)."""
    import torch
    fg = gt[:, 0].float()                                                    # [K,1,H,W]
    bg = (fg.sum(0, keepdim=True) < 0.5).float()
    return torch.cat([bg, fg], 0)


def dataset_factory(dataset, prop_net, fuse_net, device="cuda:0", mem_freq=5, core_cls=None):
    """engine_factory for `run_suite` over a real dataset: clip `spec.clip_id` is decoded (PIL) and ingested by the loader (HIP kernels when the
    dataset was built with device=...), an InferenceCore is built over it and the first frame's annotation becomes the interacted mask -
    the semi-supervised protocol BASELINE config 4 names (YouTube-VOS val: first-frame masks given).  `core_cls` is injectable for CPU tests."""
    if core_cls is None:
        from .inference_core import InferenceCore as core_cls

    def factory(spec):
        data = dataset[spec.clip_id]
        images, gt = data["rgb"].unsqueeze(0), data["gt"]
        core = core_cls(prop_net, fuse_net, images, gt.shape[0], mem_profile=0, mem_freq=mem_freq, device=device)
        return core, first_frame_mask(gt)
    return factory


def png_writer(dataset, out_dir, palette):
    """`on_clip` callback of run_suite: the clip's masks as palette PNGs under <out_dir>/<video name>/ (eval_interactive_davis.py:86-94's egress)."""
    import os
    from . import clip_io

    def on_clip(spec, masks):
        clip_io.write_palette_png(masks, palette, os.path.join(out_dir, dataset.videos[spec.clip_id]))
    return on_clip


def generator_cost(n_frames, n_objects, separation):
    """Relative cost of a clip in generator mode: one two-sided propagation over the clip per reference frame."""
    return len(range(0, n_frames, separation)) * shard.clip_cost(n_frames - 1, n_objects)


def run_generator_suite(specs, generator_factory, rank=0, world=1, separation=5, max_objects=5, sync=None, on_result=None):
    """The offline fusion-data generator (reference generate_fusion.py:68-120) over this rank's share of `specs`: for every
    `separation`-th frame of a clip, the frame's ground-truth masks of the usable objects (more than 10 x 10 pixels, at most
    `max_objects`: generate_fusion.py:83-91) are propagated to both ends of the clip (DAVIS ranges, :100-102) and the soft
    probabilities come back as uint8 maps (:109).  ``generator_factory(spec) -> (generator, gt)`` with gt [T,K,1,H,W] float
    masks (objects only) and generator a FusionGenerator over the clip.  Records: one per clip, `frames` = propagated frames."""
    parts = shard.assign_sequences([generator_cost(s.frames, s.objects, separation) for s in specs], world)
    records = []
    for i in parts[rank]:
        spec = specs[i]
        gen, gt = generator_factory(spec)
        if sync:
            sync()
        t0 = time.perf_counter()
        crc, frames, refs = 0, 0, 0
        for frame in range(0, spec.frames, separation):
            usable = [k for k in range(gt.shape[1]) if float((gt[frame, k] > 0.5).sum()) > 100][:max_objects]
            if not usable:
                continue
            gen.reset(len(usable))
            probs = gen.interact_mask(gt[frame, usable], frame, 0, spec.frames - 1)          # [K+1, T, H, W]
            out = np.asarray((probs[1:] * 255).to("cpu").numpy() if hasattr(probs, "to") else probs[1:] * 255).astype(np.uint8)
            crc = zlib.crc32(np.ascontiguousarray(out).tobytes(), crc)
            frames += spec.frames - 1
            refs += 1
            if on_result is not None:
                on_result(spec, frame, usable, out)
        if sync:
            sync()
        records.append(dict(clip=spec.clip_id, rank=rank, frames=frames, objects=spec.objects, reference_frames=refs,
                            seconds=time.perf_counter() - t0, checksum=int(crc)))
        del gen
    return records


def fit_cost_model(records):
    """Least-squares fit of the measured per-clip seconds to  frames * (a + b * objects)  (the form of shard.clip_cost, whose
    constant is b / a): returns dict(a_ms, b_ms, per_object_ratio, max_rel_residual)."""
    A = np.array([[r["frames"], r["frames"] * r["objects"]] for r in records], dtype=np.float64)
    y = np.array([r["seconds"] for r in records], dtype=np.float64)
    if len(records) < 2 or len({r["objects"] for r in records}) < 2:
        return None
    (a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
    resid = np.abs(A @ np.array([a, b]) - y) / y
    return dict(a_ms=round(float(a) * 1e3, 4), b_ms=round(float(b) * 1e3, 4), per_object_ratio=round(float(b / a), 3) if a > 0 else None,
                max_rel_residual=round(float(resid.max()), 3))


def predicted_imbalance(specs, worlds=(2, 4, 8), per_object=None):
    """Busiest rank's load / mean load of the longest-first assignment, for the cost model frames * (1 + per_object * objects)
    (per_object=None: shard.clip_cost's constant).  1.0 = perfect balance; the scaling efficiency of a sharded run is at most its
    inverse."""
    out = {}
    for w in worlds:
        if per_object is None:
            cost = [shard.clip_cost(s.frames - 1, s.objects) for s in specs]
        else:
            cost = [(s.frames - 1) * (1.0 + per_object * s.objects) for s in specs]
        parts = shard.assign_sequences(cost, w)
        loads = [sum(cost[i] for i in p) for p in parts]
        out[str(w)] = round(max(loads) / (sum(loads) / w), 4)
    return out


def summarize(all_records, n_specs=None):
    """Aggregate of the gathered records of all ranks: every clip exactly once, total propagated frames, the busiest
    rank's time (what bounds the wall clock of the sharded run) and the imbalance of the assignment."""
    clips = sorted(r["clip"] for r in all_records)
    if len(set(clips)) != len(clips) or (n_specs is not None and clips != list(range(n_specs))):
        raise RuntimeError("suite sharding error: clips processed %d, unique %d, expected %s" % (len(clips), len(set(clips)), n_specs))
    per_rank = {}
    for r in all_records:
        per_rank[r["rank"]] = per_rank.get(r["rank"], 0.0) + r["seconds"]
    frames = sum(r["frames"] for r in all_records)
    busiest = max(per_rank.values()) if per_rank else 0.0
    return dict(clips=len(clips), frames=frames, busiest_rank_seconds=busiest,
                mean_rank_seconds=sum(per_rank.values()) / max(1, len(per_rank)),
                frames_per_second=frames / busiest if busiest > 0 else 0.0,
                checksum=int(sum(r["checksum"] for r in all_records) & 0xFFFFFFFFFFFF))

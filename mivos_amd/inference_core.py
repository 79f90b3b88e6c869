"""InferenceCore for MI355X: bidirectional space-time-memory propagation + difference-aware fusion.

Drop-in for the reference's `inference_core.py:17-293` (same constructor, ``interact`` /
``update_mask_only`` / ``get_image_buffered``, same public attributes ``prob, masks, np_masks, images,
pad, k, t, h, w, nh, nw``), re-designed around the HIP engine:

  * the control flow of a pass is computed up front by ``plan_pass`` (pure Python, unit-tested
    against the reference's golden schedule) and then executed;
  * the memory bank is one pre-allocated ``[K, slots, h, w, C]`` buffer per pass; ``memorize`` writes
    its keys/values straight into the slot from the KeyValue GEMM epilogue (no copy), the reader gets
    ``bank[:, :n]`` as strides;
  * per-frame query features are cached together with the object-independent decoder skip branches,
    so re-propagation after a later interaction skips 128 GMAC/frame of encoder + skip work;
  * fusion runs all K objects as one batch (one attention launch, one FusionNet chain);
  * the final argmax over all T frames is a single launch.
"""
import functools
import os
from collections import namedtuple

import numpy as np
import torch

from . import ops
from .model.fusion_net import FusionNet
from .model.propagation.prop_net import CK, CV, PropagationNetwork
from .util.tensor_util import pad_divide_by

Step = namedtuple("Step", "ti n_read slot fuse")


def plan_frames(frames, idx, mem_freq, n_certain, fuse=False):
    """Memory schedule of a run of frames propagated from frame idx (reference do_pass, inference_core.py:165-198;
    generation/fusion_generator.py:58-78).  Returns (total_slots, steps).  Each step: frame index ``ti``, number of leading bank
    slots the reader sees (``n_read``), the slot the frame's own key/value are written to afterwards (``None`` for the last
    frame of the run) and whether the frame is fused.  Every propagated frame is written to the slot at the front (a temporary
    'previous frame' memory); the front only advances - i.e. the frame is kept - when it is at least ``mem_freq`` frames from
    the last kept one."""
    total = len(frames) // mem_freq + 1 + n_certain
    steps, front, last_kept, prev_kept = [], n_certain, idx, True
    for i, ti in enumerate(frames):
        n_read = front if prev_kept else front + 1
        slot = None
        if i != len(frames) - 1:
            slot = front
            prev_kept = abs(ti - last_kept) >= mem_freq
            if prev_kept:
                front, last_kept = front + 1, ti
        steps.append(Step(ti, n_read, slot, fuse))
    return total, steps


def plan_pass(t, interacted, idx, forward, mem_freq, n_certain):
    """Schedule of one propagation pass of InferenceCore (reference: do_pass, inference_core.py:122-200): the frames between idx
    and the closest interacted frame in that direction (or the end of the clip), fused iff such a frame exists.
    Returns (closest, total_slots, steps) - see plan_frames."""
    if forward:
        closest = min([x for x in interacted if x > idx] + [t])
        frames = list(range(idx + 1, closest))
    else:
        closest = max([x for x in interacted if x < idx] + [-1])
        frames = list(range(idx - 1, closest, -1))
    total, steps = plan_frames(frames, idx, mem_freq, n_certain, fuse=closest != t and closest != -1)
    return closest, total, steps


def _on_core_device(fn):
    """Public entry points run with the core's GPU as the current HIP device (the kernels launch on the current device's
    stream), whatever device the caller had selected."""
    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapped


class InferenceCore:
    """
    images       - unpadded, normalised CPU tensor [1,T,3,H,W]
    mem_profile  - 0: everything resident in HBM (default; a 1080p x 1000-frame clip is ~70 GB of the
                   288 GB).  1-3 keep images (and for 2-3 results) on the host with bounded caches,
                   like the reference (inference_core.py:44-63); accuracy is unaffected.
    mem_freq     - every mem_freq-th propagated frame is kept in the memory bank
    """

    def __init__(self, prop_net: PropagationNetwork, fuse_net: FusionNet, images, num_objects,
                 mem_profile=0, mem_freq=5, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ops.MivosHipError("InferenceCore needs an MI355X device; mivos_amd has no CPU execution path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        # .to() of a network that already lives on the device keeps its compiled plan (packed weights); parameters that
        # autograd saw change since the plan was built (optimiser steps, p.copy_, re-assignment) are detected here, once per
        # clip; writes through p.data are invisible to that check and need invalidate_plan() (model/plan_cache.py)
        self.prop_net = prop_net.to(self.device)
        self.prop_net.refresh_plan_if_stale()
        if fuse_net is not None:
            self.fuse_net = fuse_net.to(self.device)
            self.fuse_net.refresh_plan_if_stale()
        self.mem_profile, self.mem_freq = mem_profile, mem_freq
        self.data_dev = self.device if mem_profile == 0 else torch.device("cpu")
        self.result_dev = self.device if mem_profile in (0, 1) else torch.device("cpu")
        self.q_buf_size = {0: 105, 1: 105, 2: 3}.get(mem_profile, 1)
        self.i_buf_size = {0: -1, 1: 105, 2: 3}.get(mem_profile, 1)

        self.t = images.shape[1]
        self.h, self.w = images.shape[-2:]
        self.k = num_objects
        self.images, self.pad = pad_divide_by(images, 16, images.shape[-2:])
        self.nh, self.nw = self.images.shape[-2:]
        self.images = self.images.to(self.data_dev)
        # the default precision carries operands as fp16 hi + lo pairs: inputs beyond the fp16 range would turn into inf / NaN inside
        # the first convolution (and ReLU would silently turn those into zeros) - refuse them here (one reduction per clip)
        # (`not (amax < 65504)`: a NaN frame must be refused too - NaN compares false against everything)
        if ops.CONV_PRECISION == "f16x3" and not (max(abs(float(v)) for v in self.images.aminmax()) < 65504.0):
            raise ops.MivosHipError("InferenceCore: |images| >= 65504 is outside the fp16 range of the f16x3 operands (INTEGRATION.md 'Limits'); "
                                    "normalise the frames (dataset/range_transform.py) or set ops.CONV_PRECISION = 'f32'")
        self.kh, self.kw = self.nh // 16, self.nw // 16

        self.masks = torch.zeros((self.t, 1, self.nh, self.nw), dtype=torch.uint8, device=self.result_dev)
        self.np_masks = np.zeros((self.t, self.h, self.w), dtype=np.uint8)
        self.prob = torch.zeros((self.k + 1, self.t, 1, self.nh, self.nw), dtype=torch.float32, device=self.result_dev)
        self.prob[0] = 1e-7

        self.query_buf, self.image_buf = {}, {}
        self._lookahead = set()                      # frames encoded ahead of their use (not yet consumed by a pass)
        self.interacted = set()
        self._certain_k = self._certain_v = None     # [K, n, h, w, C] rows per memory position
        self.propagated_frames = 0                   # do_pass iterations so far (the bench metric)
        self._fuse_stream = self._pass_stream = None
        self._range = ops.new_range_status(self.device)      # this core's fp16-range flag (ops.range_status): raised by ITS launches, read by it alone

    # ---- reference-shaped views of the certain memory -------------------------------------
    @property
    def certain_mem_k(self):
        return None if self._certain_k is None else self._certain_k.permute(0, 4, 1, 2, 3)

    @property
    def certain_mem_v(self):
        return None if self._certain_v is None else self._certain_v.permute(0, 4, 1, 2, 3)

    # ---- caches -----------------------------------------------------------------------------
    @_on_core_device
    def get_image_buffered(self, idx):
        if self.data_dev == self.device:
            return self.images[:, idx]
        if idx not in self.image_buf:
            if len(self.image_buf) > self.i_buf_size:
                self.image_buf = {}
            self.image_buf[idx] = self.images[:, idx].to(self.device)
        return self.image_buf[idx]

    # frames encoded together on a cache miss (when the cache has room).  Bigger batches fill the chip better (16 / 32: +1.2 % / +2 %
    # in steady state) but every frame encoded ahead and not consumed inside a measurement window is charged to it: with 8 the driver's
    # 20-step window encoded 24 frames (192.9 frames/s); 10 divides it and the 69-frame pass nearly evenly: 202.2 on the same box, 196.1
    # vs 195.3 over two whole sessions (profiles/r04b_knob_ab.txt; 20 measures the same).  bench.py drops the look-ahead at t0 whatever
    # the size, so no timed frame is ever encoded before the clock starts.
    QUERY_BATCH = int(os.environ.get("MIVOS_QUERY_BATCH", "10"))

    def _encode(self, todo):
        if len(todo) == 1:
            return [self.prop_net.encode_query(self.get_image_buffered(todo[0]))]
        return self.prop_net.encode_query_batch(torch.cat([self.get_image_buffered(t) for t in todo], 0))

    def _query(self, idx, upcoming=()):
        """Cached query features of frame idx.  On a miss the next not-yet-cached frames of the running pass
        (`upcoming`, in processing order) are encoded in the same batch: the features are state independent,
        so this is the reference's lazy cache (:110-120) filled a few frames ahead.
        (Encoding the NEXT batch on a side stream while the current one is consumed was built and measured twice: round 3 +0.8 % on the full
        config-3 session; round 6, same box, one clip in flight +1 % (201.9 vs 200.1 frames/s), two clips in flight -17 % (187 vs 225: a fifth
        and sixth active stream on HIP's four hardware queues), on a lowest-priority HIP stream -5 % / +0 % (profiles/r06e_query_prefetch_ab.txt).
        Not kept.)"""
        q = self.query_buf.get(idx)
        if q is None:
            if len(self.query_buf) > self.q_buf_size:
                self.query_buf, self._lookahead = {}, set()
            room = self.q_buf_size + 1 - len(self.query_buf)
            todo = [idx] + [t for t in upcoming if t != idx and t not in self.query_buf]
            todo = todo[:max(1, min(self.QUERY_BATCH, room))]
            for t, qt in zip(todo, self._encode(todo)):
                self.query_buf[t] = qt
            self._lookahead.update(todo[1:])
            q = self.query_buf[idx]
        self._lookahead.discard(idx)
        return q

    def drop_lookahead(self):
        """Forget query features that were encoded ahead of their frame's turn (benchmark hygiene: a timed region that
        starts here contains the encoding work of every frame it propagates).  Returns how many were dropped."""
        n = len(self._lookahead)
        for t in self._lookahead:
            self.query_buf.pop(t, None)
        self._lookahead = set()
        return n

    @_on_core_device
    def get_query_kv_buffered(self, idx):
        return self._query(idx).as_reference_tuple()

    # Fusion of frame ti only feeds self.prob[:, ti]; the propagation chain (memorize -> next frame's read) never looks at it
    # (reference inference_core.py:186-197: the frame is memorised from the UNFUSED mask).  With everything resident in HBM the
    # fusion launches therefore go to a second HIP stream behind an event and run beside the next memorize / decode launches:
    # FusionNet is HBM-bound, the encoder GEMMs are matrix-core bound.  Same kernels, same arithmetic, same results.
    FUSE_ON_SIDE_STREAM = os.environ.get("MIVOS_FUSE_SIDE_STREAM", "1") != "0"
    # Where the side stream's start event is recorded: behind `aggregate` (the fusion kernels of frame t then run beside memorize(t)) or behind memorize(t)
    # (beside frame t + 1's read and decoder).  Same kernels either way.  Measured same box (profiles/r06c_fold_ab.txt): ONE clip in flight 194.4 vs 198.0
    # frames/s (the HBM-bound fusion kernels and the memory encoder's HBM-bound 1x1 layers get in each other's way), TWO clips in flight 223.4 vs 200.9
    # (the other clip's launches fill the encoder's holes anyway).  "auto": beside memorize exactly when the caller says other streams share the chip.
    FUSE_BESIDE_MEMORIZE = os.environ.get("MIVOS_FUSE_BESIDE_MEMORIZE", "auto")

    def _fuse_async(self, closest, idx, ti, out, key_k, q, pending):
        main = torch.cuda.current_stream()
        if self._fuse_stream is None:
            self._fuse_stream = ops.side_stream(self.device, "fuse")      # shared by the cores that run under this stream (warm allocator pool, workspaces)
        side = self._fuse_stream
        ready = torch.cuda.Event()
        ready.record(main)
        out.record_stream(side)                       # allocated on the main stream, last read on the side stream
        with torch.cuda.stream(side):
            side.wait_event(ready)
            self.prob[:, ti] = self.fuse_one_frame(closest, idx, ti, self.prob[:, ti], out, key_k, q.k16)
        pending.append((out, q))                      # keep the operands alive until the streams have joined

    def _join_fusion(self, pending):
        if pending:
            torch.cuda.current_stream().wait_stream(self._fuse_stream)
            del pending[:]

    # ---- one propagation pass ---------------------------------------------------------------
    def _pass_steps(self, plan, key_k, key_v, idx, step_cb=None):
        """Generator over the propagated frames of one pass (`plan` = plan_pass's result): everything a frame needs is enqueued
        on the CURRENT HIP stream, then the generator yields - nothing in a step waits for the GPU, so a driver can advance
        several passes (of this core: `interact`, or of other cores: eval_suite.run_suite(lanes=2)) in turn, each under its own
        stream, and the chip overlaps their under-filled launches (one 480p frame of five objects fills a quarter of the CUs
        at the 30 x 54 layers).  The fusion branch is joined into the pass's stream when the generator finishes."""
        closest, total, steps = plan
        nc = self._certain_k.shape[1]
        K, kh, kw = self.k, self.kh, self.kw
        keys = torch.empty((K, total, kh, kw, CK), dtype=torch.float32, device=self.device)
        values = torch.empty((K, total, kh, kw, CV), dtype=torch.float32, device=self.device)
        keys[:, :nc], values[:, :nc] = self._certain_k, self._certain_v
        # the affinity kernel streams keys pre-split into fp16 hi/lo pairs (ops.split_keys): a second bank of the same size,
        # every slot converted once when it is written
        # (f16x3 only: the exact-fp32 verification mode streams the fp32 rows themselves)
        ksplit = torch.empty_like(keys) if ops.affinity_precision() == "f16x3" else None
        if ksplit is not None:
            ops.split_keys(keys[:, :nc], ksplit[:, :nc])
        hw = kh * kw
        pending = []
        for si, st in enumerate(steps):
            with ops.range_status(self._range):              # (per step: never held across the yield)
                q = self._query(st.ti, upcoming=[s2.ti for s2 in steps[si + 1:si + self.QUERY_BATCH]])
                prob_k = self.prop_net.segment(keys[:, :st.n_read].reshape(K, st.n_read * hw, CK),
                                               values[:, :st.n_read].reshape(K, st.n_read * hw, CV), q,
                                               keys_split=None if ksplit is None else ksplit[:, :st.n_read].reshape(K, st.n_read * hw, CK))
                out = ops.aggregate(prob_k.unsqueeze(1), keep_bg=True)            # [K+1,1,nh,nw]
                side_fuse = st.fuse and self.FUSE_ON_SIDE_STREAM and self.result_dev == self.device
                beside = side_fuse and (ops.CHIP_SHARE > 1 if self.FUSE_BESIDE_MEMORIZE == "auto" else self.FUSE_BESIDE_MEMORIZE not in ("0", 0, False))
                if beside:
                    # enqueued BEFORE memorize: the side stream's start event then sits right behind `aggregate`, and the fusion kernels run beside
                    # this frame's memory encoder (under-filled 30 x 54 layers) instead of beside the NEXT frame's read + decoder
                    self._fuse_async(closest, idx, st.ti, out, key_k, q, pending)
                if st.slot is not None:
                    self.prop_net.memorize_into(self.get_image_buffered(st.ti), out[1:],
                                                key_out=keys[:, st.slot], val_out=values[:, st.slot])
                    if ksplit is not None:
                        ops.split_keys(keys[:, st.slot], ksplit[:, st.slot])
                if beside:
                    pass
                elif side_fuse:
                    self._fuse_async(closest, idx, st.ti, out, key_k, q, pending)
                else:
                    if st.fuse:
                        out = self.fuse_one_frame(closest, idx, st.ti, self.prob[:, st.ti], out, key_k, q.k16)
                    self.prob[:, st.ti] = out.to(self.result_dev)
            self.propagated_frames += 1
            if step_cb is not None:
                step_cb()
            yield st.ti
        self._join_fusion(pending)

    def do_pass(self, key_k, key_v, idx, forward=True, step_cb=None):
        """key_k: keys of the interacted frame, rows layout [K, h*w, 128] (used by the fusion attention)."""
        plan = plan_pass(self.t, self.interacted, idx, forward, self.mem_freq, self._certain_k.shape[1])
        for _ in self._pass_steps(plan, key_k, key_v, idx, step_cb=step_cb):
            pass
        return plan[0]

    # The forward and the backward pass of an interaction never exchange data (reference inference_core.py:255-256: two do_pass calls
    # over disjoint frame ranges that only READ the certain memory): with everything resident in HBM they advance in turn, frame
    # by frame, on two HIP streams.  Same kernels, same inputs, bit-identical results (round 6: the split-K slicing of the convolutions no
    # longer depends on how many streams share the chip); MIVOS_CONCURRENT_PASSES=0 runs them one after the other.
    CONCURRENT_PASSES = os.environ.get("MIVOS_CONCURRENT_PASSES", "1") != "0"
    PASS_CHIP_SHARE = 2     # what the kernels are told while the two passes are in flight (ops.chip_share: launch geometry only, results do not depend on it)

    def _run_passes(self, rows, key_v, idx, step_cb=None):
        nc = self._certain_k.shape[1]
        plans = [plan_pass(self.t, self.interacted, idx, fwd, self.mem_freq, nc) for fwd in (True, False)]
        plans = [p for p in plans if p[2]]
        # (a clip longer than the query cache may flush it mid-interaction: a cached feature tensor freed on one stream while the other
        # pass still reads it - such clips keep the sequential order)
        if len(plans) < 2 or not self.CONCURRENT_PASSES or self.result_dev != self.device or self.t > self.q_buf_size:
            for p in plans:
                for _ in self._pass_steps(p, rows, key_v, idx, step_cb=step_cb):
                    pass
            return
        main = torch.cuda.current_stream()
        if self._pass_stream is None:
            self._pass_stream = ops.side_stream(self.device, "pass")
        side = self._pass_stream
        side.wait_stream(main)                                   # the interacted frame's keys / values / difference maps
        lanes = [(main, self._pass_steps(plans[0], rows, key_v, idx, step_cb=step_cb)),
                 (side, self._pass_steps(plans[1], rows, key_v, idx, step_cb=step_cb))]
        for t in (rows, key_v):
            t.record_stream(side)
        with ops.chip_share(self.PASS_CHIP_SHARE * ops.CHIP_SHARE):
            while lanes:
                for lane in list(lanes):
                    with torch.cuda.stream(lane[0]):
                        if next(lane[1], None) is None:
                            lanes.remove(lane)
        main.wait_stream(side)

    @_on_core_device
    def fuse_logits(self, tc, tr, ti, prev_mask, curr_mask, mk16, qk16):
        """FusionNet's logits [K,nh,nw,1] for frame ti (reference inference_core.py:202-215 up to the sigmoid), all K objects in
        one batch: aligned difference maps (get_attention) + FusionNet on (frame, previous result, new propagation, maps,
        normalised distances).  mk16: [K, h*w, 128] rows or the reference's [K,128,1,h,w]; qk16: NHWC [1,h,w,128] or the
        reference's [1,128,h,w]."""
        assert tc < ti < tr or tr < ti < tc
        K, P = self.k, self.nh * self.nw
        nc, nr = abs(tc - ti) / abs(tc - tr), abs(tr - ti) / abs(tc - tr)
        if mk16.dim() == 5:          # called the reference's way: NCHW-shaped key / query tensors
            mk16 = mk16.permute(0, 2, 3, 4, 1).reshape(K, -1, CK)
            qk16 = qk16.permute(0, 2, 3, 1)
        hw = self.kh * self.kw
        low = self.prop_net.attention_lowres(mk16, self._pos16, self._neg16, qk16.reshape(hw, CK))
        attn = ops.resize_bilinear(low.view(K * 2, self.kh, self.kw), self.nh, self.nw)      # [2K, nh, nw]
        prev = prev_mask.to(self.device)
        curr = curr_mask.to(self.device)
        im = self.get_image_buffered(ti).contiguous()
        prev_k, curr_k = prev[1:], curr[1:]
        return self.fuse_net.run_planes((im, 0), (prev_k, prev_k.stride(0)), (curr_k, curr_k.stride(0)),
                                        (attn, 2 * P), (nc, nr), K)                         # [K,nh,nw,1]

    @_on_core_device
    def fuse_one_frame(self, tc, tr, ti, prev_mask, curr_mask, mk16, qk16):
        """Difference-aware fusion of the previous result with the new propagation for frame ti
        (reference inference_core.py:202-217): sigmoid of fuse_logits, aggregated with the soft background."""
        w = ops.sigmoid(self.fuse_logits(tc, tr, ti, prev_mask, curr_mask, mk16, qk16))
        return ops.aggregate(w.view(self.k, 1, self.nh, self.nw), keep_bg=True)

    def _prepare_diff(self, mask, old):
        """Positive / negative difference of the new (padded, one-hot) mask of the interacted frame against its previous
        probabilities (reference :236-238), full resolution and area-pooled to the key grid (the same for every fused frame)."""
        K = self.k
        self.pos_mask_diff, self.neg_mask_diff = ops.mask_diff(mask, old)
        self._pos16 = ops.area_pool16(self.pos_mask_diff[1:].reshape(K, self.nh, self.nw)).view(K, -1)
        self._neg16 = ops.area_pool16(self.neg_mask_diff[1:].reshape(K, self.nh, self.nw)).view(K, -1)

    # ---- public entry points ------------------------------------------------------------------
    def _begin_interaction(self, mask, idx, total_cb=None):
        """Everything of `interact` before the passes (reference :219-253): register the frame, difference maps against the previous
        result, memorise the interacted frame into the certain memory.  Returns (key rows [K, h*w, 128], values) of that frame."""
        self.interacted.add(idx)
        self._range.zero_()                                   # a flag left by an interaction that aborted must not fail this one
        mask = mask.to(self.device).float()
        mask, _ = pad_divide_by(mask, 16, mask.shape[-2:])
        mask = mask.contiguous()
        K = self.k
        self._prepare_diff(mask, self.prob[:, idx].to(self.device))
        self.prob[:, idx] = mask.to(self.result_dev)

        with ops.range_status(self._range):
            key_k, key_v = self.prop_net.memorize_into(self.get_image_buffered(idx), mask[1:])   # [K,h,w,C]
        key_k5, key_v5 = key_k.unsqueeze(1), key_v.unsqueeze(1)
        if self._certain_k is None:
            self._certain_k, self._certain_v = key_k5, key_v5
        else:
            self._certain_k = torch.cat([self._certain_k, key_k5], 1)
            self._certain_v = torch.cat([self._certain_v, key_v5], 1)

        if total_cb is not None:
            front = min([ti for ti in self.interacted if ti > idx] + [self.t])
            back = max([ti for ti in self.interacted if ti < idx] + [-1])
            if front - back - 2 > 0:
                total_cb(front - back - 2)
        return key_k.reshape(K, self.kh * self.kw, CK), key_v

    @_on_core_device
    def interact(self, mask, idx, total_cb=None, step_cb=None):
        """mask: one-hot [K+1,1,H,W] (background first) of frame idx.  Propagates both ways from idx,
        fusing with earlier results between interacted frames.  Returns uint8 [T,H,W]."""
        rows, key_v = self._begin_interaction(mask, idx, total_cb)
        self._run_passes(rows, key_v, idx, step_cb=step_cb)
        return self._refresh_masks()

    def interact_steps(self, mask, idx, total_cb=None, step_cb=None):
        """`interact` as a generator: yields after every propagated frame with that frame's work enqueued on the CURRENT HIP stream
        and nothing waited for; when it is exhausted the result is in `np_masks` (and is the StopIteration value).  A driver that
        advances the generators of several cores in turn, each under its own stream (eval_suite.run_suite(lanes=2)), overlaps
        their launches on the chip.  The two passes of one interaction run one after the other here."""
        # (the core's device is made current PER STEP, never across a yield: interleaved generators of cores on different devices would
        # otherwise restore each other's "previous" device)
        with torch.cuda.device(self.device):
            rows, key_v = self._begin_interaction(mask, idx, total_cb)
            nc = self._certain_k.shape[1]
        for fwd in (True, False):
            plan = plan_pass(self.t, self.interacted, idx, fwd, self.mem_freq, nc)
            if not plan[2]:
                continue
            steps = self._pass_steps(plan, rows, key_v, idx, step_cb=step_cb)
            while True:
                with torch.cuda.device(self.device):
                    ti = next(steps, None)
                if ti is None:
                    break
                yield ti
        with torch.cuda.device(self.device):
            return self._refresh_masks()

    REFRESH_CHUNK_BYTES = 1 << 30     # host-resident results (mem_profile 2/3): probabilities visit the GPU in chunks

    def _refresh_masks(self):
        """argmax over objects for every frame, crop the padding, copy to the host.  Results resident in HBM: one launch
        over all T frames.  Results on the host (mem_profile 2/3, the reference's low-memory profiles): bounded chunks of
        frames are uploaded, so the GPU footprint stays O(chunk) like the reference's per-frame loop (:259-260)."""
        l, r, t, b = self.pad
        P = self.nh * self.nw
        # (Rounds 2-3 probed the probabilities for NaN here.  That probe could never fire: an activation beyond the fp16 range of the
        # f16x3 operands becomes inf / NaN inside a convolution, but every ReLU (fmaxf) on the way - each bottleneck's output, the
        # decoder's `pred` input - and the clamp of aggregate_wbg return finite numbers for NaN, so the corruption is silent by the
        # time it reaches `prob`.  The guards that exist: the input check of __init__ and, since round 5, the convolution epilogues' range
        # flag read below.)
        out = self._argmax_and_copy(l, r, t, b, P)
        if ops.CONV_PRECISION == "f16x3":          # the epilogues' fp16-range guard (round 5): an overflow INSIDE the network is no longer silent
            ops.check_activation_range(self._range)  # (this core's own word: the passes' streams were joined into the current one)
        return out

    PINNED_RESULT_BYTES = 256 << 20

    @staticmethod
    def _to_host_u8(view):
        """uint8 device view [T, h, w] -> a numpy array of its own (the interaction's result, 29 MB for 70 frames of 480p): cropped into a dense
        tensor on the GPU, ONE copy into page-locked host memory (torch's caching host allocator: the block is recycled once the array is dropped),
        one stream synchronisation.  The pageable `.cpu().numpy().astype(uint8)` of rounds 1-5 cost two more host-side passes over the array
        (~5 ms per interaction with the GPU idle)."""
        dense = view.contiguous()
        if dense.numel() > InferenceCore.PINNED_RESULT_BYTES:      # very long clips (config 5: 2 GB of masks): not worth page-locking that much host memory
            return dense.cpu().numpy()
        host = torch.empty(dense.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(dense, non_blocking=True)
        torch.cuda.current_stream(dense.device).synchronize()
        return host.numpy()

    def _argmax_and_copy(self, l, r, t, b, P):
        if self.prob.device == self.device:
            m = ops.argmax_u8(self.prob.view(self.k + 1, self.t * P)).view(self.t, 1, self.nh, self.nw)
            self.masks = m
            self.np_masks = self._to_host_u8(m[:, 0, t:self.nh - b, l:self.nw - r])
            return self.np_masks
        step = max(1, self.REFRESH_CHUNK_BYTES // ((self.k + 1) * P * 4))
        for t0 in range(0, self.t, step):
            t1 = min(self.t, t0 + step)
            chunk = self.prob[:, t0:t1].to(self.device).contiguous()
            m = ops.argmax_u8(chunk.view(self.k + 1, (t1 - t0) * P)).view(t1 - t0, 1, self.nh, self.nw)
            self.masks[t0:t1] = m.to(self.result_dev)
        self.np_masks = self.masks[:, 0, t:self.nh - b, l:self.nw - r].cpu().numpy().astype(np.uint8)
        return self.np_masks

    @_on_core_device
    def update_mask_only(self, prob_mask, idx):
        """Interaction without propagation (reference :273-293): prob_mask [K+1,1,nh,nw] (padded)."""
        prob_mask = prob_mask.to(self.device).float().contiguous()
        m = ops.argmax_u8(prob_mask.view(prob_mask.shape[0], -1)).view(1, self.nh, self.nw)
        self.masks[idx] = m.to(self.result_dev)
        l, r, t, b = self.pad
        self.np_masks[idx] = m[0, t:self.nh - b, l:self.nw - r].cpu().numpy().astype(np.uint8)
        return self.np_masks

"""FusionGenerator for MI355X - the offline generator of FusionNet training data (reference `generation/fusion_generator.py:12-101`,
driven by `generate_fusion.py:68-120`): for a reference frame of a clip, propagate its (given) masks to both range limits
WITHOUT fusion and return the soft probabilities of every frame.  Same constructor, ``reset`` / ``interact_mask`` / ``do_pass`` /
``get_im`` / ``get_query_buf`` as the reference; built on the engine's fast path:

  * the bank is one pre-allocated ``[K, slots, h, w, C]`` buffer per pass (the reference grows it with ``torch.cat`` on every kept
    frame), ``memorize`` writes its slot from the KeyValue GEMM epilogue, the reader sees ``bank[:, :n]`` as strides;
  * query features (and the decoder's object-independent skip branches) are cached per frame ACROSS reference frames: the
    reference re-encodes every frame for every reference frame of the clip (`get_query_buf` has no cache), although the features
    do not depend on the propagation state - with `generate_fusion.py`'s separation of 5 frames on a DAVIS clip that is a
    14-fold redundancy;
  * missing frames of a pass are encoded a batch at a time like InferenceCore does.
"""
import os

import torch

from .. import ops
from ..inference_core import plan_frames
from ..model.aggregate import aggregate_wbg
from ..model.propagation.prop_net import CK, CV, PropagationNetwork
from ..util.tensor_util import pad_divide_by


class FusionGenerator:
    QUERY_BATCH = 8

    def __init__(self, prop_net: PropagationNetwork, images, mem_freq):
        self.mem_freq = mem_freq
        self.t = images.shape[1]
        self.h, self.w = images.shape[-2:]
        images, self.pad = pad_divide_by(images, 16, images.shape[-2:])
        self.nh, self.nw = images.shape[-2:]
        self.images = images
        self.device = images.device
        if self.device.type != "cuda":
            raise ops.MivosHipError("FusionGenerator needs the clip on an MI355X device (images.cuda()); mivos_amd has no CPU execution path")
        self.prop_net = prop_net.to(self.device)
        self.prop_net.refresh_plan_if_stale()
        self.kh, self.kw = self.nh // 16, self.nw // 16
        self.query_buf = {}
        self.propagated_frames = 0
        self._pass_stream = None
        self._range = ops.new_range_status(self.device)      # this generator's fp16-range flag (ops.range_status)

    def reset(self, k):
        self.k = k
        self.prob = torch.zeros((k + 1, self.t, 1, self.nh, self.nw), dtype=torch.float32, device=self.device)

    def get_im(self, idx):
        return self.images[:, idx]

    def _query(self, idx, upcoming=()):
        q = self.query_buf.get(idx)
        if q is None:
            todo = ([idx] + [t for t in upcoming if t != idx and t not in self.query_buf])[:self.QUERY_BATCH]
            if len(todo) == 1:
                self.query_buf[idx] = self.prop_net.encode_query(self.get_im(idx))
            else:
                for t, qt in zip(todo, self.prop_net.encode_query_batch(torch.cat([self.get_im(t) for t in todo], 0))):
                    self.query_buf[t] = qt
            q = self.query_buf[idx]
        return q

    def get_query_buf(self, idx):
        with ops.on_device(self.device):
            return self._query(idx).as_reference_tuple()

    def _pass_steps(self, key_k, key_v, idx, left_limit, right_limit, forward):
        """One pass as a generator: every `next()` enqueues one propagated frame on the current HIP stream and waits for nothing (the
        form InferenceCore._pass_steps has; `interact_mask` advances the forward and the backward pass in turn on two streams)."""
        if key_k.dim() == 5 and key_k.shape[1] == CK:
            key_k, key_v = key_k[:, :, 0].permute(0, 2, 3, 1), key_v[:, :, 0].permute(0, 2, 3, 1)
        frames = list(range(idx + 1, right_limit + 1)) if forward else list(range(idx - 1, left_limit - 1, -1))
        if not frames:
            return
        total, steps = plan_frames(frames, idx, self.mem_freq, 1)
        K, kh, kw, hw = self.k, self.kh, self.kw, self.kh * self.kw
        keys = torch.empty((K, total, kh, kw, CK), dtype=torch.float32, device=self.device)
        values = torch.empty((K, total, kh, kw, CV), dtype=torch.float32, device=self.device)
        keys[:, 0], values[:, 0] = key_k, key_v
        ksplit = torch.empty_like(keys) if ops.affinity_precision() == "f16x3" else None
        if ksplit is not None:
            ops.split_keys(keys[:, :1], ksplit[:, :1])
        for si, st in enumerate(steps):
            q = self._query(st.ti, upcoming=[s2.ti for s2 in steps[si + 1:si + self.QUERY_BATCH]])
            prob_k = self.prop_net.segment(keys[:, :st.n_read].reshape(K, st.n_read * hw, CK), values[:, :st.n_read].reshape(K, st.n_read * hw, CV), q,
                                           keys_split=None if ksplit is None else ksplit[:, :st.n_read].reshape(K, st.n_read * hw, CK))
            out = ops.aggregate(prob_k.unsqueeze(1), keep_bg=True)
            self.prob[:, st.ti] = out
            if st.slot is not None:
                self.prop_net.memorize_into(self.get_im(st.ti), out[1:], key_out=keys[:, st.slot], val_out=values[:, st.slot])
                if ksplit is not None:
                    ops.split_keys(keys[:, st.slot], ksplit[:, st.slot])
            self.propagated_frames += 1
            yield st.ti

    def do_pass(self, key_k, key_v, idx, left_limit, right_limit, forward=True):
        """key_k / key_v: keys / values of the annotated frame, rows layout [K,h,w,C] (or the reference's [K,C,1,h,w])."""
        for _ in self._pass_steps(key_k, key_v, idx, left_limit, right_limit, forward):
            pass

    # the two passes from a reference frame are independent (fusion_generator.py:97-98: two do_pass calls over disjoint frames): they advance in
    # turn on two HIP streams, like the two passes of InferenceCore.interact (MIVOS_CONCURRENT_PASSES=0: one after the other)
    CONCURRENT_PASSES = os.environ.get("MIVOS_CONCURRENT_PASSES", "1") != "0"
    PASS_CHIP_SHARE = 2

    def _run_passes(self, key_k, key_v, idx, left_limit, right_limit):
        both = idx < right_limit and idx > left_limit
        if not (both and self.CONCURRENT_PASSES):
            self.do_pass(key_k, key_v, idx, left_limit, right_limit, True)
            self.do_pass(key_k, key_v, idx, left_limit, right_limit, False)
            return
        main = torch.cuda.current_stream()
        if self._pass_stream is None:
            self._pass_stream = ops.side_stream(self.device, "pass")
        side = self._pass_stream
        side.wait_stream(main)
        for t in (key_k, key_v):
            t.record_stream(side)
        lanes = [(main, self._pass_steps(key_k, key_v, idx, left_limit, right_limit, True)),
                 (side, self._pass_steps(key_k, key_v, idx, left_limit, right_limit, False))]
        with ops.chip_share(self.PASS_CHIP_SHARE * ops.CHIP_SHARE):
            while lanes:
                for lane in list(lanes):
                    with torch.cuda.stream(lane[0]):
                        if next(lane[1], None) is None:
                            lanes.remove(lane)
        main.wait_stream(side)

    def interact_mask(self, mask, idx, left_limit, right_limit):
        """mask [K,1,H,W]: the objects' masks of frame idx (generate_fusion.py:104) -> probabilities [K+1, T, H, W] of the frames
        left_limit .. right_limit (the others keep reset()'s zeros)."""
        with ops.on_device(self.device), torch.no_grad():
            self._range.zero_()
            with ops.range_status(self._range):               # (no yield leaves this function: the passes' generators are advanced inside it)
                mask = mask.to(self.device).float()
                mask, _ = pad_divide_by(mask, 16, mask.shape[-2:])
                mask = aggregate_wbg(mask.contiguous(), keep_bg=True)
                self.prob[:, idx] = mask
                key_k, key_v = self.prop_net.memorize_into(self.get_im(idx), mask[1:])
                self._run_passes(key_k, key_v, idx, left_limit, right_limit)
            if ops.CONV_PRECISION == "f16x3":                 # an overflow inside the network must not be silent here either (one 4-byte read per call)
                ops.check_activation_range(self._range)
            l, r, t, b = self.pad
            return self.prob[:, :, 0, t:self.nh - b, l:self.nw - r]

"""DAVISProcessor for MI355X — the junction between the DAVIS interactive track and InferenceCore (reference
`davis_processor.py:12-92`): scribbles -> S2M -> hard aggregated mask -> interact / update_mask_only on the DAVIS schedule.

Same constructor, ``to_mask`` / ``interact`` and bookkeeping (``davis_schedule = [2, 5, 7]``) as the reference.  The reference
rasterises the scribble paths with ``davisinteractive.utils.scribbles.scribbles2mask`` and dilates with ``cv2.dilate``; neither
package is needed here: the dilation is a HIP kernel and ``interact_scribble_mask`` takes the rasterised label map directly
(``interact(scribble_dict)`` imports davisinteractive lazily for the rasterisation only).
"""
import numpy as np
import torch

from . import ops
from .inference_core import InferenceCore
from .model.aggregate import aggregate_wbg
from .util.tensor_util import pad_divide_by


class DAVISProcessor:
    def __init__(self, prop_net, fuse_net, s2m_net, images, num_objects, device="cuda:0"):
        self.device = torch.device(device)
        self.s2m_net = s2m_net.to(self.device)
        self.s2m_net.refresh_plan_if_stale()                    # packed weights follow parameters changed since the last clip
        _, self.pad = pad_divide_by(images[:, :1], 16, images.shape[-2:])
        self.t = images.shape[1]
        self.h, self.w = images.shape[-2:]                      # true dimensions (the reference overwrites them with the padded ones,
        self.k = num_objects                                    # davis_processor.py:28-32; its scribbles are rasterised on THAT canvas: to_mask)
        self.interacted_count = 0
        self.davis_schedule = [2, 5, 7]
        self.processor = InferenceCore(prop_net, fuse_net, images, num_objects, mem_profile=0, device=device)
        self.nh, self.nw = self.processor.nh, self.processor.nw

    def mask_from_scribble_mask(self, scr_mask, idx):
        """scr_mask: int array [h,w] of the interacted frame, -1 = no scribble, 0 = background scribble, j = object j
        (scribbles2mask's output).  davis_processor.py:52-70 -> hard aggregated mask [K+1,1,nh,nw]."""
        scr = torch.as_tensor(np.asarray(scr_mask)).to(self.device)
        # [h,w]: a label map of the true frame (callers that rasterise themselves); [nh,nw]: the padded canvas the reference's
        # to_mask rasterises on (see to_mask)
        assert tuple(scr.shape) in ((self.h, self.w), (self.nh, self.nw))
        sh, sw = scr.shape
        with ops.on_device(self.device):
            K = self.k
            ids = torch.arange(1, K + 1, device=self.device).view(K, 1, 1)
            pos = (scr[None] == ids).float()                                     # [K,h,w]
            neg = ((scr[None] != ids) & (scr[None] != -1)).float()
            rs = ops.dilate3x3(torch.cat([pos, neg], 0)).view(2, K, sh, sw)      # cv2.dilate, 3x3 ones
            rs, _ = pad_divide_by(rs, 16, rs.shape[-2:])                         # padded AFTER the dilation, like the reference
            frame = self.processor.get_image_buffered(idx)                       # [1,3,nh,nw]
            cur = self.processor.masks[idx].to(self.device)                      # [1,nh,nw] uint8
            # one S2M forward for all K objects (the reference calls the network once per object, davis_processor.py:62-68): sample
            # ki = cat([frame, current mask of object ki (hard: S2M is trained with such), its positive / negative scribbles])
            cur_k = (cur[None] == ids.view(K, 1, 1, 1)).float()                      # [K,1,nh,nw]
            inputs = torch.cat([frame.expand(K, -1, -1, -1), cur_k, rs[0].unsqueeze(1), rs[1].unsqueeze(1)], 1)
            if getattr(self, "_s2m_range", None) is None:
                self._s2m_range = ops.new_range_status(self.device)
            with ops.range_status(self._s2m_range):                                  # the S2M convolutions' fp16-range flag: this processor's own word
                mask = ops.sigmoid(self.s2m_net(inputs))
            out = aggregate_wbg(mask, keep_bg=True, hard=True)
            if ops.CONV_PRECISION == "f16x3":
                ops.check_activation_range(self._s2m_range)                          # (one 4-byte read per interaction)
            return out

    def to_mask(self, scribble):
        """The reference's entry point: a DAVIS scribble dict (davis_processor.py:38-50)."""
        from davisinteractive.utils.scribbles import scribbles2mask      # rasterisation only; not needed by the rest
        all_scr = scribble["scribbles"]
        idx = 0
        for idx, s in enumerate(all_scr):
            if len(s) != 0:
                scribble["scribbles"] = [s]
                break
        # The reference pads the clip BEFORE it stores "the true dimensions" (davis_processor.py:20-32: self.h, self.w are the padded
        # ones), so its scribble paths - normalised to [0, 1] over the true frame - are rasterised over the PADDED canvas (854 -> 864
        # columns stretches a stroke by 1.2 %).  Results have to match the reference's, so the same canvas is used here.
        scr_mask = scribbles2mask(scribble, (self.nh, self.nw))[0]
        return self.mask_from_scribble_mask(scr_mask, idx), idx

    def _advance(self, mask, idx):
        if self.interacted_count == self.davis_schedule[0]:
            self.davis_schedule = self.davis_schedule[1:]                 # finish the instant-interaction loop for this frame
            next_interact = None
            out_masks = self.processor.interact(mask, idx)
        else:
            next_interact = [idx]
            out_masks = self.processor.update_mask_only(mask, idx)
        self.interacted_count += 1
        return out_masks, next_interact, idx                              # np_masks are already cropped to the true size

    def interact(self, scribble):
        mask, idx = self.to_mask(scribble)
        return self._advance(mask, idx)

    def scribble_canvas(self):
        """(rows, columns) on which a caller of interact_scribble_mask must rasterise normalised scribble paths to reproduce
        interact(): the PADDED frame (nh, nw) - see to_mask."""
        return self.nh, self.nw

    def interact_scribble_mask(self, scr_mask, idx):
        """interact() for callers that rasterise the scribbles themselves.  For the reference's results the label map has to be
        rasterised on `scribble_canvas()` = the padded (nh, nw) frame (what to_mask does, like davis_processor.py:38-50); a map of the
        true (h, w) frame is accepted too and is taken as is - its strokes then sit where the caller put them, up to 1.2 % (854 -> 864
        columns) away from where interact() would put the same normalised paths."""
        return self._advance(self.mask_from_scribble_mask(scr_mask, idx), idx)

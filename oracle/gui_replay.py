"""PyQt-free replay of how the reference's GUI drives its processor (TEST INFRASTRUCTURE ONLY).

`/root/reference/interactive_gui.py` cannot run here (PyQt5, cv2, a display), but what it does to `InferenceCore` is a handful of
statements.  This module restates exactly those statements - each method cites the GUI lines it follows - so that the same scripted
session can be replayed on the unmodified reference (CPU; `oracle/make_golden_gui.py` -> tests/golden/gui_small.npz), on the CPU
oracle and on the MI355X engine under `torch.cuda.amp.autocast` (interactive_gui.py:990 wraps the whole application in it).  The
interaction objects of the GUI (scribble -> S2M, click -> f-BRS, free drawing) are outside the hot path; their product, the soft
`interacted_mask` [K+1,1,nh,nw], is synthesised here (`draw`: a free-hand box, interact/interaction.py's FreeInteraction writes
ones / zeros into a clone of the current probabilities the same way).

What is exercised beyond `interact`: `prob[:, i].clone()` as an interaction's starting point, `update_mask_only` after every edit
with its result read back through `np_masks[i]`, in-place edits of the processor's public buffers (`masks[i].zero_()`,
`np_masks[i].fill(0)`), the `current_mask` alias of `np_masks`, the progress callbacks, and the read-only attributes the local-refinement
mode takes (`prob[1:, i]`, `images[:, i]`, `pad`)."""
import numpy as np
import torch


class GuiReplay:
    def __init__(self, processor, num_objects):
        self.processor = processor                      # interactive_gui.py:57-58
        self.k = num_objects
        self.cursur = 0                                 # (sic) the GUI's name for the current frame
        self.interacted_mask = None
        self.debug_mask = None
        self.current_mask = np.zeros_like(processor.np_masks)      # :267 np.zeros((num_frames, height, width), uint8)
        self.progress = []                              # what the progress bar was told
        self.events = []                                # (name, copy of current_mask) after every handler

    def _log(self, name):
        self.events.append((name, self.current_mask.copy()))

    # ---- timeline ------------------------------------------------------------------------------------------------------------
    def goto(self, frame):
        self.cursur = frame

    # ---- an edit of the current frame ------------------------------------------------------------------------------------------
    def start_from_current(self):
        """interactive_gui.py:616 / :626 (undo of the whole interaction), and the starting point of every free-hand interaction:
        the frame's current probabilities, cloned."""
        self.interacted_mask = self.processor.prob[:, self.cursur].clone()

    def draw(self, obj, y0, y1, x0, x1):
        """A free-hand stroke for object `obj` (0 = erase to background) as the padded box [y0:y1, x0:x1]: that object's channel
        becomes 1, every other channel 0, inside the box."""
        m = self.interacted_mask
        m[:, :, y0:y1, x0:x1] = 0
        m[obj, :, y0:y1, x0:x1] = 1
        self.update_interacted_mask("draw")

    def set_mask(self, onehot_padded):
        """A complete mask for the frame (the --masks option / a scribble's S2M result)."""
        self.interacted_mask = onehot_padded.to(self.processor.prob.device).float().clone()
        self.update_interacted_mask("set_mask")

    def update_interacted_mask(self, name="update"):
        """interactive_gui.py:889-897 (global mode): update_mask_only, then the frame's row of np_masks into current_mask."""
        self.processor.update_mask_only(self.interacted_mask, self.cursur)
        self.current_mask[self.cursur] = self.processor.np_masks[self.cursur]
        self._log(name)

    def debug_pressed(self):
        """interactive_gui.py:955-960: swap the edit with the stored one and show it."""
        self.debug_mask, self.interacted_mask = self.interacted_mask, self.debug_mask
        self.processor.update_mask_only(self.interacted_mask, self.cursur)
        self.current_mask[self.cursur] = self.processor.np_masks[self.cursur]
        self._log("debug")

    # ---- buttons ---------------------------------------------------------------------------------------------------------------
    def on_run(self):
        """interactive_gui.py:542-556: propagate from the current frame; `current_mask` becomes the array interact returns."""
        assert self.interacted_mask is not None
        self.current_mask = self.processor.interact(self.interacted_mask, self.cursur, self.progress_total_cb, self.progress_step_cb)
        self.interacted_mask = None
        self._log("run")

    def on_reset(self):
        """interactive_gui.py:636-642: clear the frame's result IN PLACE in the processor's buffers (prob is left alone: the next
        interaction still needs the mask difference)."""
        self.processor.masks[self.cursur].zero_()
        self.processor.np_masks[self.cursur].fill(0)
        self.current_mask[self.cursur].fill(0)
        self.interacted_mask = None
        self._log("reset")

    def local_mode_inputs(self):
        """interactive_gui.py:662-680 (on_finish_local) / the local-refinement set-up: what it reads from the processor."""
        return dict(prev_soft_mask=self.processor.prob[1:, self.cursur].detach().float().cpu().numpy().copy(),
                    image=self.processor.images[:, self.cursur].detach().float().cpu().numpy().copy(), pad=tuple(int(p) for p in self.processor.pad))

    def progress_total_cb(self, total):                 # :537-540
        self.progress.append(("total", int(total)))

    def progress_step_cb(self):                         # :530-535
        self.progress.append(("step",))


SESSION = dict(t=7, h=100, w=141, k=2, seed=23, mem_freq=2, top_k=20, conditioning="synthetic.CLOSED_LOOP_CONDITIONING")


def session_states():
    """The session's weights: the seeded synthetic state dicts with the closed-loop conditioning (mivos_amd/util/synthetic.py) - on them the
    reference's own fp32 and fp64 runs of this session agree to IoU >= 0.9999 (0-2 pixels) after every handler, so IoU >= 0.999 against the
    golden is a fair bar for a second implementation (on the unconditioned weights the reference is at 0.9989 from itself)."""
    from mivos_amd.util import synthetic
    from oracle import weights as Wt
    return synthetic.condition_state(Wt.make_prop_state(0), **synthetic.CLOSED_LOOP_CONDITIONING), Wt.make_fuse_state(0)


def scripted_session(processor, gt, cfg=SESSION):
    """The scripted GUI session of the golden vector.  gt: one-hot masks [T,K+1,1,h,w] of the clip (synthetic).  Returns the GuiReplay
    (events, progress) plus the local-mode inputs read at the end."""
    from oracle.stm_oracle import pad_divide_by
    g = GuiReplay(processor, cfg["k"])
    first, _ = pad_divide_by(gt[0].float(), 16)
    g.goto(0); g.set_mask(first); g.on_run()                                     # annotate frame 0, propagate forward
    last, _ = pad_divide_by(gt[cfg["t"] - 1].float(), 16)
    g.goto(cfg["t"] - 1); g.set_mask(last); g.on_run()                           # annotate the last frame: backward pass, fused
    g.goto(3); g.start_from_current(); g.draw(1, 20, 60, 30, 90)                  # free-hand correction in the middle ...
    g.debug_mask = processor.prob[:, 3].clone()
    g.debug_pressed(); g.debug_pressed()                                          # ... compared with the old result, twice (back to the edit)
    g.draw(2, 50, 80, 70, 120)
    g.on_run()                                                                    # both passes, both fused
    g.goto(2); g.on_reset()                                                       # throw frame 2's result away ...
    g.start_from_current(); g.draw(0, 0, 40, 0, 50); g.on_run()                   # ... erase a corner to background and propagate again
    g.goto(5); g.start_from_current(); g.update_interacted_mask("undo_all")       # undo of a whole interaction: the frame's own probabilities
    return g, g.local_mode_inputs()

"""DavisInteractiveSession of the stand-in `davisinteractive` (see the package docstring): the loop protocol of
`eval_interactive_davis.py:72-108` driven by a scripted robot.

Samples: every sequence of <davis_root>/ImageSets/2017/val.txt twice (two "users"), sequence by sequence.  A sample has
`max_nb_interactions` interactions.  The frame of an interaction is the single candidate the method asked for through
``submit_masks(.., next_scribble_frame_candidates=[idx])`` (the reference does after every update_mask_only), otherwise the next
entry of a fixed per-user list.  The scribbles of interaction n on frame f: one horizontal stroke per object through the GT mask
of that object (row = its centroid row + a small n-dependent offset, clipped to the object) and one background stroke along the
row with the fewest foreground pixels - functions of the annotation only.
`MIVOS_STUB_LOG=<file.npz>`: every submitted mask array is stored there at session exit (interaction log for parity tests)."""
import json
import os

import numpy as np
from PIL import Image


class DavisInteractiveSession:
    def __init__(self, host="localhost", user_key=None, davis_root=None, subset="val", shuffle=False, max_time=None, max_nb_interactions=8,
                 metric_to_optimize="J", report_save_dir=None):
        self.root = davis_root
        with open(os.path.join(davis_root, "ImageSets", "2017", "val.txt")) as f:
            seqs = [l.strip() for l in f if l.strip()]
        self.samples = [(s, u) for s in seqs for u in range(2)]
        self.max_inter = max_nb_interactions
        self.sample_i, self.inter_i, self.started = -1, 0, False
        self.candidates, self.log, self.report = None, [], []
        self._gt = {}

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        path = os.environ.get("MIVOS_STUB_LOG")
        if path and exc[0] is None:
            np.savez_compressed(path, **{f"sub_{i:03d}_{s}_u{u}_n{n}_f{f}": m for i, (s, u, n, f, m) in enumerate(self.log)})
        return False

    def _annotations(self, seq):
        if seq not in self._gt:
            d = os.path.join(self.root, "Annotations", "480p", seq)
            self._gt[seq] = np.stack([np.array(Image.open(os.path.join(d, f)).convert("P"), dtype=np.uint8) for f in sorted(os.listdir(d))])
        return self._gt[seq]

    def next(self):
        if not self.started or self.inter_i >= self.max_inter:
            self.sample_i += 1
            self.inter_i, self.candidates, self.new_sample = 0, None, True
            self.started = True
        return self.sample_i < len(self.samples)

    def _frame(self, gt, user):
        if self.candidates:
            return int(self.candidates[0])
        t = gt.shape[0]
        order = [0, t - 1, t // 2] if user == 0 else [t - 1, 1, t // 2]
        return order[(self.inter_i // 3) % 3]

    def get_scribbles(self, only_last=False):
        seq, user = self.samples[self.sample_i]
        gt = self._annotations(seq)
        t, h, w = gt.shape
        f = self._frame(gt, user)
        lab = gt[f]
        lines = []
        for k in [int(v) for v in np.unique(gt[0]) if v != 0]:
            ys, xs = np.nonzero(lab == k)
            if len(ys) == 0:
                continue
            rows = np.unique(ys)
            y = int(rows[min(len(rows) - 1, max(0, len(rows) // 2 + (self.inter_i % 3 - 1) * max(1, len(rows) // 6)))])
            xr = xs[ys == y]
            x0, x1 = int(xr.min()), int(xr.max())
            x0, x1 = x0 + (x1 - x0) // 5, x1 - (x1 - x0) // 5
            lines.append(dict(path=[[x / (w - 1), y / (h - 1)] for x in range(x0, x1 + 1, 3)] or [[x0 / (w - 1), y / (h - 1)]], object_id=k, start_time=0, end_time=0))
        fg = (lab != 0).sum(1)
        yb = int(np.argmin(fg + np.abs(np.arange(h) - (h // 4 + 7 * (self.inter_i % 3))) * 1e-3))
        free = np.nonzero(lab[yb] == 0)[0]
        if len(free):
            lines.append(dict(path=[[int(x) / (w - 1), yb / (h - 1)] for x in free[::4]], object_id=0, start_time=0, end_time=0))
        scribbles = dict(sequence=seq, scribbles=[lines if i == f else [] for i in range(t)])
        new = self.new_sample
        self.new_sample = False
        self._cur = (seq, user, f)
        return seq, scribbles, new

    def submit_masks(self, pred_masks, next_scribble_frame_candidates=None):
        seq, user, f = self._cur
        gt = self._annotations(seq)
        pm = np.asarray(pred_masks)
        assert pm.shape == gt.shape, (pm.shape, gt.shape)
        ks = [int(v) for v in np.unique(gt[0]) if v != 0]
        j = float(np.mean([((pm == k) & (gt == k)).sum() / max(1, ((pm == k) | (gt == k)).sum()) for k in ks]))
        self.report.append(dict(sequence=seq, user=user, interaction=self.inter_i, frame=f, J=j))
        self.log.append((seq, user, self.inter_i, f, pm.astype(np.uint8).copy()))
        self.candidates = list(next_scribble_frame_candidates) if next_scribble_frame_candidates else None
        self.inter_i += 1

    def get_report(self):
        return self.report

    def get_global_summary(self, save_file=None):
        js = [r["J"] for r in self.report]
        summary = dict(submissions=len(js), mean_J=float(np.mean(js)) if js else 0.0, final_J={f"{r['sequence']}/u{r['user']}": r["J"] for r in self.report
                                                                                                if r["interaction"] == self.max_inter - 1})
        if save_file:
            with open(save_file, "w") as f:
                json.dump(summary, f)
        return summary

"""Golden vector of the GUI's call pattern (`/root/reference/interactive_gui.py`), produced by the UNMODIFIED reference InferenceCore on
PyTorch-CPU driven by `oracle/gui_replay.py` (the PyQt-free restatement of the GUI's handlers).

TEST INFRASTRUCTURE ONLY (this container).  Writes ``tests/golden/gui_small.npz``: `current_mask` after every handler of the scripted
session (oracle/gui_replay.py::scripted_session), the progress-bar calls, the processor's final `masks` / `prob`, and what the
local-refinement mode reads.  The CPU oracle must reproduce all of it bit for bit (checked here and in tests/test_oracle_golden.py).

    python -m oracle.make_golden_gui
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gui_replay as G  # noqa: E402
from oracle import ref_loader  # noqa: E402
from oracle import stm_oracle as O  # noqa: E402
from oracle import weights as Wt  # noqa: E402


def pack(g, local, core):
    out = {"config": json.dumps(G.SESSION), "event_names": np.array([n for n, _ in g.events]),
           "progress": np.array([p[1] if p[0] == "total" else -1 for p in g.progress], dtype=np.int64)}
    for i, (_, m) in enumerate(g.events):
        out[f"current_mask_{i}"] = m
    out["final_masks"] = core.masks.cpu().numpy().copy()
    out["final_prob"] = core.prob.float().cpu().numpy().copy()
    out["final_np_masks"] = core.np_masks.copy()
    out["local_prev_soft_mask"], out["local_image"], out["local_pad"] = local["prev_soft_mask"], local["image"], np.asarray(local["pad"])
    return out


def main():
    torch.set_grad_enabled(False)
    cfg = G.SESSION
    ref, prop, fuse = ref_loader.build_reference_networks(top_k=cfg["top_k"])
    sd, fsd = G.session_states()
    prop.load_state_dict(sd)
    fuse.load_state_dict(fsd)
    images, gt = O.synthetic_clip(cfg["t"], cfg["h"], cfg["w"], cfg["k"], cfg["seed"])
    core = ref["inference_core"].InferenceCore(prop, fuse, images, cfg["k"], mem_profile=0, mem_freq=cfg["mem_freq"], device="cpu")
    g, local = G.scripted_session(core, gt)
    out = pack(g, local, core)
    ocore = O.OracleCore(sd, fsd, images, cfg["k"], mem_freq=cfg["mem_freq"], top_k=cfg["top_k"])
    og, olocal = G.scripted_session(ocore, gt)
    oout = pack(og, olocal, ocore)
    for k in out:
        same = np.array_equal(out[k], oout[k])
        if not same:
            print("oracle differs from the reference on", k)
    # admission of the fixture: the reference's arithmetic in fp64 (oracle, bit-identical to the reference in fp32) against its fp32 run
    from mivos_amd.util.tensor_util import compute_np_iou
    o64 = O.OracleCore(sd, fsd, images, cfg["k"], mem_freq=cfg["mem_freq"], top_k=cfg["top_k"], dtype=torch.float64)
    g64, _ = G.scripted_session(o64, gt)
    self_iou = [float(np.mean([compute_np_iou(a == j, b == j) for j in range(1, cfg["k"] + 1)])) for (_, a), (_, b) in zip(g.events, g64.events)]
    out["self_iou_fp32_vs_fp64"] = np.asarray(self_iou)
    print("reference fp32 vs fp64 IoU per event:", " ".join(f"{x:.5f}" for x in self_iou))
    assert min(self_iou) >= 0.9995, "fixture not well conditioned"
    print("events:", [n for n, _ in g.events])
    print("progress:", out["progress"].tolist())
    path = os.path.join(ROOT, "tests", "golden", "gui_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

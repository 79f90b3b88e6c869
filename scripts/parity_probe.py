"""Closed-loop parity of the bench clip vs the CPU oracle, per frame / object, for both conv back-ends."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from mivos_amd import ops
from mivos_amd.inference_core import InferenceCore
from mivos_amd.model.fusion_net import FusionNet
from mivos_amd.model.propagation.prop_net import PropagationNetwork
from mivos_amd.util import synthetic
from oracle import stm_oracle as O
K, F = 5, int(os.environ.get("FRAMES", "4"))
torch.set_num_threads(32)
sd, fsd = synthetic.make_prop_state(0), synthetic.make_fuse_state(0)
CT = int(os.environ.get("CLIP_T", F + 1))
images, gt = synthetic.synthetic_clip(CT, 480, 854, K, seed=int(os.environ.get("SEED", "100")))
images, gt = images[:, :F + 1], gt[:F + 1]
O.TOPK_GAP = []
oc = O.OracleCore(sd, fsd, images, K, mem_freq=5, top_k=50)
ref = oc.interact(gt[0], 0)
print("oracle min top-k margin", min(O.TOPK_GAP))
for prec in ("f32", "f16x3"):
    ops.CONV_PRECISION = prec
    prop, fuse = PropagationNetwork(top_k=50), FusionNet()
    prop.load_state_dict(sd); fuse.load_state_dict(fsd)
    core = InferenceCore(prop, fuse, images, K, mem_freq=5, device="cuda:0")
    out = core.interact(gt[0], 0)
    dp = (core.prob.cpu() - oc.prob).abs()
    for ti in range(1, F + 1):
        ious = [((out[ti] == j) & (ref[ti] == j)).sum() / max(1, ((out[ti] == j) | (ref[ti] == j)).sum()) for j in range(1, K + 1)]
        print(prec, "frame", ti, "mismatch px", int((out[ti] != ref[ti]).sum()), "max|dprob| %.2e" % float(dp[:, ti].max()),
              "IoU", " ".join("%.5f" % x for x in ious), "areas", [int((ref[ti] == j).sum()) for j in range(1, K + 1)])

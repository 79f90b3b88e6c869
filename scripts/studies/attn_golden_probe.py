"""AttentionReadNetwork golden (tests/test_gpu_engine.py::test_attention_read_network_golden) as a stand-alone probe: prints the two max |d| for the package
found under argv[1] (a tree root), so that two trees can be compared on one box.  python scripts/studies/attn_golden_probe.py <tree root> <repo root>"""
import os
import sys

import numpy as np
import torch

tree, repo = sys.argv[1], sys.argv[2]
sys.path.insert(0, tree)
torch.set_grad_enabled(False)
from mivos_amd.model.attn_network import AttentionReadNetwork  # noqa: E402
from mivos_amd.util import synthetic  # noqa: E402

with np.load(os.path.join(repo, "tests", "golden", "attn_small.npz")) as z:
    g = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
sd = synthetic.make_prop_state(0)
net = AttentionReadNetwork()
net.load_state_dict({k: v for k, v in sd.items() if not k.startswith("decoder.")})
net = net.to("cuda:0").eval()
for rep in range(3):
    a1, a2 = net(*(g[n].to("cuda:0") for n in ("an_image", "an_m11", "an_m21", "an_m12", "an_m22", "an_query")))
    d = (a2.cpu() - g["an_out2"]).abs()
    print(tree, "max|d|", float((a1.cpu() - g["an_out1"]).abs().max()), float(d.max()), "q999", float(d.flatten().kthvalue(int(d.numel() * 0.999)).values), "n>2e-5", int((d > 2e-5).sum()), "of", d.numel())

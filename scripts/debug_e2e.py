import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_grad_enabled(False)
from mivos_amd import ops
from mivos_amd.inference_core import InferenceCore
from mivos_amd.model.fusion_net import FusionNet
from mivos_amd.model.propagation.prop_net import PropagationNetwork
from oracle import stm_oracle as O, weights as Wt
DEV = "cuda:0"
sd, fsd = Wt.make_prop_state(0), Wt.make_fuse_state(0)
prop = PropagationNetwork(top_k=20); prop.load_state_dict(sd); prop.to(DEV)
fuse = FusionNet(); fuse.load_state_dict(fsd); fuse.to(DEV)
z = np.load("tests/golden/e2e_small.npz"); c = json.loads(str(z["config"]))
images, gt = O.synthetic_clip(c["t"], c["h"], c["w"], c["k"], c["seed"])
core = InferenceCore(prop, fuse, images, c["k"], mem_freq=c["mem_freq"], device=DEV)
o32 = O.OracleCore(sd, fsd, images, c["k"], mem_freq=c["mem_freq"], top_k=c["top_k"])
o64 = O.OracleCore(sd, fsd, images, c["k"], mem_freq=c["mem_freq"], top_k=c["top_k"], dtype=torch.float64)

# single-layer / encoder error vs fp64 truth
f0 = images[:, 0]
k, v = prop.memorize_into(f0.to(DEV), gt[0, 1:].to(DEV))
k32, v32 = O.memorize(sd, f0, gt[0, 1:]); k64, v64 = O.memorize(sd, f0.double(), gt[0, 1:].double())
kk = k.permute(0, 3, 1, 2).cpu().double()
print("memorize keys: engine-vs-fp64 %.3e   cpu32-vs-fp64 %.3e  (max |k| %.2f)" % (float((kk - k64[:, :, 0]).abs().max()), float((k32.double() - k64).abs().max()), float(k64.abs().max())))
q = prop.encode_query(f0.to(DEV)); q32 = O.get_query_values(sd, f0); q64 = O.get_query_values(sd, f0.double())
for n, a, b, cc in zip(("f16", "f8", "f4", "k16", "v16"), (q.f16, q.f8, q.f4, q.k16, q.v16), q32, q64):
    print("query %s: engine-vs-fp64 %.3e   cpu32-vs-fp64 %.3e" % (n, float((a.permute(0, 3, 1, 2).cpu().double() - cc).abs().max()), float((b.double() - cc).abs().max())))

seg = {}
orig = prop.segment
def rec(keys, values, q, logits=False):
    out = orig(keys, values, q, logits=True)
    seg[len(seg)] = out.cpu()
    return ops.sigmoid(out) if not logits else out
prop.segment = rec
core.interact(gt[0], 0); o32.interact(gt[0], 0); o64.interact(gt[0], 0)
for i, ti in enumerate(sorted(o32.logits)):
    l64 = o64.logits[ti][:, 0]
    print("frame", ti, "logit: engine-vs-fp64 %.2e  cpu32-vs-fp64 %.2e  engine-vs-cpu32 %.2e" % (
        float((seg[i].double() - l64).abs().max()), float((o32.logits[ti][:, 0].double() - l64).abs().max()), float((seg[i] - o32.logits[ti][:, 0]).abs().max())),
        "mask mism eng/64 %d  32/64 %d" % (int((core.np_masks[ti] != o64.np_masks[ti]).sum()), int((o32.np_masks[ti] != o64.np_masks[ti]).sum())))

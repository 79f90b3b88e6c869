"""Stage-by-stage comparison of the HIP engine against the CPU oracle (debug aid, run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from mivos_amd import ops
from mivos_amd.model.propagation.prop_net import PropagationNetwork, QueryFeatures, CK, CV
from mivos_amd.model.propagation.modules import run_resblock, run_up_branch, run_skip_branch
from oracle import stm_oracle as O, weights as Wt
import torch.nn.functional as F

DEV = "cuda:0"
sd = Wt.make_prop_state(0)
prop = PropagationNetwork(top_k=20); prop.load_state_dict(sd); prop.to(DEV)
images, gt = O.synthetic_clip(3, 128, 160, 2, seed=5)
f0, f1 = images[:, 0], images[:, 1]

def d(name, got, ref):
    got = got.cpu()
    print(f"{name:28s} max|d| {float((got-ref).abs().max()):.3e}  ref max {float(ref.abs().max()):.3e}")

def nchw(x): return x.permute(0, 3, 1, 2)

ok, ov = O.memorize(sd, f0, gt[0, 1:])
oq = O.get_query_values(sd, f1)
k, v = prop.memorize_into(f0.to(DEV), gt[0, 1:].to(DEV))
q = prop.encode_query(f1.to(DEV))
d("mem key", nchw(k), ok[:, :, 0]); d("mem val", nchw(v), ov[:, :, 0])
for n, a, b in zip(("f16", "f8", "f4", "k16", "v16"), (q.f16, q.f8, q.f4, q.k16, q.v16), oq):
    d("query " + n, nchw(a), b)
K, h, w = 2, 8, 10
om = torch.cat([O.memory_read(ok[i:i+1], ov[i:i+1], oq[3], 20) for i in range(K)], 0)
om4 = torch.cat([om, oq[4].expand(K, -1, -1, -1)], 1)
m4 = torch.empty((K, h, w, 2 * CV), device=DEV)
ops.memory_read(k.reshape(K, h * w, CK), v.reshape(K, h * w, CV), q.k16.view(h * w, CK), 20, out=m4.view(K, h * w, 2 * CV)[:, :, :CV])
m4[..., CV:] = q.v16
d("m4", nchw(m4), om4)
dec = prop.plan()["dec"]
x = run_resblock(dec["compress"], m4)
ox = O.res_block(sd, "decoder.compress.", om4)
d("compress", nchw(x), ox)
s8 = run_skip_branch(dec["up_16_8"], q.f8)
os8 = O.res_block(sd, "decoder.up_16_8.skip_conv2.", O._conv(sd, "decoder.up_16_8.skip_conv1", oq[1], pad=1))
d("skip8", nchw(s8), os8)
u = ops.upsample2x_add(s8, x)
ou = os8 + F.interpolate(ox, scale_factor=2, mode="bilinear", align_corners=False)
d("up2x+add", nchw(u), ou)
x = run_resblock(dec["up_16_8"][2], u)
ox = O.res_block(sd, "decoder.up_16_8.out_conv.", ou)
d("up_16_8", nchw(x), ox)
s4 = run_skip_branch(dec["up_8_4"], q.f4)
os4 = O.res_block(sd, "decoder.up_8_4.skip_conv2.", O._conv(sd, "decoder.up_8_4.skip_conv1", oq[2], pad=1))
d("skip4", nchw(s4), os4)
x = run_up_branch(dec["up_8_4"], s4, x)
ox = O.upsample_block(sd, "decoder.up_8_4.", oq[2], ox)
d("up_8_4", nchw(x), ox)
lo = ops.conv(x, dec["pred"], relu_in=True)
olo = O._conv(sd, "decoder.pred", F.relu(ox), pad=1)
d("pred", nchw(lo), olo)

"""FusionNet training step on the MI355X (reference model/fusion_model.py:54-131, model/losses.py, train.py:96-124)
against the reference's own outputs (tests/golden/train_small.npz, oracle/make_golden_train.py), the CPU training oracle and
torch autograd.  Needs an MI355X; every compute call goes through the C ABI."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mivos_amd import ops
from oracle import train_oracle as TO
from oracle import weights as Wt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("cx,cg,N,H,W", [(32, 32, 2, 37, 61), (16, 32, 3, 16, 30), (32, 1, 2, 33, 47), (32, 32, 8, 64, 64)])
def test_wgrad3x3_vs_autograd(cx, cg, N, H, W):
    """mivos_fusion_wgrad3x3 (exact fp32 MFMA, deterministic) vs the fp64 weight / bias gradient of F.conv2d."""
    g = torch.Generator().manual_seed(cx * 100 + cg + H)
    x = torch.randn(N, cx, H, W, generator=g)
    gy = torch.randn(N, cg, H, W, generator=g)
    w = torch.zeros(cg, cx, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cg, dtype=torch.float64, requires_grad=True)
    with torch.enable_grad():
        (F.conv2d(x.double(), w, b, padding=1) * gy.double()).sum().backward()
    xd, gd = x.permute(0, 2, 3, 1).contiguous().to(DEV), gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw, db = ops.fusion_wgrad3x3(xd, gd)
    dw2, db2 = ops.fusion_wgrad3x3(xd, gd)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                       # deterministic summation order
    ref = w.grad.permute(0, 2, 3, 1)                                           # OIHW -> OHWI
    scale = float(ref.abs().max())
    assert dw.shape == (cg, 3, 3, cx) and float((dw.cpu().double() - ref).abs().max()) < 2e-5 * scale
    assert float((db.cpu().double() - b.grad).abs().max()) < 2e-5 * max(1.0, float(b.grad.abs().max()))


def _loss_case(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    z1, z2 = torch.randn(B, H * W, generator=g) * 4, torch.randn(B, H * W, generator=g) * 4
    z1[0, :50], z2[0, 50:100] = 30.0, -30.0                                    # saturated sigmoids: the clamp of aggregate_wbg_channel
    selector = torch.tensor([[1.0, 1.0]] * (B - 1) + [[1.0, 0.0]])
    cls = torch.randint(0, 3, (B, H * W), generator=g)
    cls[B - 1] = cls[B - 1].clamp(max=1)                                       # no second object in the last sample
    return z1, z2, selector, cls


def _loss_reference(z1, z2, selector, cls, frac):
    """The reference's arithmetic (fusion_model.py:84-87, aggregate.py:39-53, losses.py:21-63) in torch with autograd - in fp32 like
    the reference: the clamp bound 1 - 1e-7 is not an fp32 number (it rounds to 1 - 2^-23), so saturated logits are 15.94 in fp32 and
    16.12 in fp64."""
    from oracle import stm_oracle as O
    B, P = z1.shape
    a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    with torch.enable_grad():
        prob = torch.stack([torch.sigmoid(a), torch.sigmoid(b)], 1) * selector.unsqueeze(2)      # [B,2,P]
        logits, mask = O.aggregate_wbg_channel(prob.unsqueeze(3), True)
        logits, mask = logits[..., 0], mask[..., 0]
        total, per_pixel = 0, []
        for j in range(B):
            lg = logits[j:j + 1] if selector[j, 1] > 0.5 else logits[j:j + 1, :2]
            raw = F.cross_entropy(lg, cls[j:j + 1], reduction="none").view(-1)
            per_pixel.append(raw.detach())
            if frac is None:
                total = total + raw.mean() / B
            else:
                total = total + torch.topk(raw, int(P * frac), sorted=False)[0].mean() / B
        total.backward()
    return logits.detach(), mask.detach(), torch.stack(per_pixel), float(total), a.grad, b.grad


@pytest.mark.parametrize("frac", [None, 0.575, 0.15])
def test_loss_kernels_vs_autograd(frac):
    """mivos_fusion_loss / _kth_loss / _loss_grad vs fp64 torch autograd through sigmoid x selector, aggregate_wbg_channel, the
    (bootstrapped) cross-entropy; incl. saturated pixels (zero gradient through the clamp) and a sample without second object."""
    B, H, W = 3, 24, 40
    P = H * W
    z1, z2, selector, cls = _loss_case(B, H, W, 5)
    logits_r, mask_r, loss_r, total_r, g1, g2 = _loss_reference(z1, z2, selector, cls, frac)
    zd1, zd2, sd, cd = z1.to(DEV), z2.to(DEV), selector.to(DEV), cls.to(torch.int32).to(DEV)
    logits, mask, loss = ops.fusion_loss(zd1, zd2, sd, cd)
    assert float((logits.cpu() - logits_r).abs().max()) < 2e-5 and float((mask.cpu() - mask_r).abs().max()) < 1e-6
    assert float((loss.cpu() - loss_r).abs().max()) < 2e-5
    if frac is None:
        wsel = torch.tensor([[-float("inf"), 1.0 / (P * B), 0.0]] * B, device=DEV)
        total = float(loss.double().sum(1).div(P).sum() / B)
    else:
        k = int(P * frac)
        sel = ops.fusion_kth_loss(loss, torch.full((B,), k, dtype=torch.int32, device=DEV))
        top = torch.topk(loss, k, dim=1)[0]
        assert torch.equal(sel[:, 0], top[:, -1])                              # exact k-th largest
        assert torch.equal(sel[:, 1], (loss > sel[:, :1]).sum(1).float()) and torch.equal(sel[:, 3], (loss == sel[:, :1]).sum(1).float())
        assert float((sel[:, 2].double() - (loss * (loss > sel[:, :1])).double().sum(1)).abs().max()) < 1e-2
        wsel = torch.stack([sel[:, 0], torch.full((B,), 1.0 / (k * B), device=DEV), (k - sel[:, 1]) / sel[:, 3] / (k * B)], 1)
        total = float(((sel[:, 2] + (k - sel[:, 1]) * sel[:, 0]) / k).double().sum() / B)
    assert abs(total - total_r) < 1e-5 * max(1.0, abs(total_r))
    dz1, dz2 = ops.fusion_loss_grad(zd1, zd2, sd, cd, loss, wsel)
    s = max(float(g1.abs().max()), float(g2.abs().max()))
    assert float((dz1.cpu() - g1).abs().max()) < 1e-4 * s and float((dz2.cpu() - g2).abs().max()) < 1e-4 * s
    assert float(dz2[B - 1].abs().max()) == 0.0                                # selector 0: no gradient into the second object


def test_adam_step_vs_torch():
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(39905, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-4, weight_decay=1e-7)
    p, m, v = p0.to(DEV), torch.zeros(39905, device=DEV), torch.zeros(39905, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(39905, generator=g) * 10.0 ** (step - 2)
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, grad.to(DEV), m, v, 1e-4, (0.9, 0.999), 1e-8, 1e-7, step)
        assert float((p.cpu() - ref.detach()).abs().max()) < 5e-7              # one ulp of the parameters (|p| up to 8); updates are ~1e-4
    st = opt.state[ref]
    assert float((m.cpu() - st["exp_avg"]).abs().max()) < 2e-6 * float(st["exp_avg"].abs().max())
    assert float((v.cpu() - st["exp_avg_sq"]).abs().max()) < 2e-6 * float(st["exp_avg_sq"].abs().max())


def _model(para=None, **kw):
    from mivos_amd.model.fusion_model import FusionModel
    para = para or dict(lr=1e-4, steps=[80], gamma=0.1, iterations=100)
    model = FusionModel(para, distributed=False, **kw)
    sd = Wt.make_prop_state(0)
    model.net.load_state_dict(Wt.make_fuse_state(0))
    model.prop_net.load_state_dict({k: v for k, v in sd.items() if not k.startswith("decoder.")}, strict=False)
    return model


def test_do_pass_matches_the_reference_golden(golden_dir):
    """FusionModel.do_pass on the batch of tests/golden/train_small.npz: attention maps, logits / mask, total loss, EVERY parameter's
    gradient (1e-4 of the layer's largest gradient) and the parameters after the Adam step, vs the unmodified reference."""
    with np.load(os.path.join(golden_dir, "train_small.npz")) as z:
        g = {k: z[k] for k in z.files}
    cfg = json.loads(str(g["config"]))
    data = {k[3:]: T(v) for k, v in g.items() if k.startswith("in.")}
    for it in cfg["its"]:
        tag = f"it{it}."
        model = _model(dict(lr=cfg["lr"], steps=[80], gamma=0.1, iterations=cfg["iterations"]))
        before = model.flat.clone()
        out = model.do_pass(dict(data), it)
        assert float((out["attn1"].cpu() - T(g[tag + "attn1"])).abs().max()) < 1e-4 and float((out["attn2"].cpu() - T(g[tag + "attn2"])).abs().max()) < 1e-4
        dl = float((out["logits"].cpu() - T(g[tag + "logits"])).abs().max())
        assert dl < 2e-3 and float((out["mask"].cpu() - T(g[tag + "mask"])).abs().max()) < 5e-4     # logits span +-16 (clamped probabilities)
        tl, tr = float(out["losses"]["total_loss"]), float(g[tag + "total_loss"])
        assert abs(tl - tr) < 2e-4 * max(1.0, abs(tr)) and abs(out["losses"]["p"] - float(g[tag + "p"])) < 1e-12
        off, worst = 0, 0.0
        for n, p in model.net.named_parameters():
            ref_g, ref_new = T(g[tag + "grad." + n]), T(g[tag + "new." + n])
            got_g = model.grad[off:off + p.numel()].view_as(p).cpu()
            rel = float((got_g - ref_g).abs().max()) / max(float(ref_g.abs().max()), 1e-12)
            worst = max(worst, rel)
            assert rel < 1e-3, (it, n, rel)
            # Adam's first step moves every parameter by ~lr * sign(g): compare where the gradient is not at rounding level
            clear = ref_g.abs() > 1e-4 * float(ref_g.abs().max())
            assert float((p.detach().cpu() - ref_new)[clear].abs().max()) < 2e-6, (it, n)
            off += p.numel()
        print(f"it {it}: max|dlogit| {dl:.2e}  loss {tl:.6f} vs {tr:.6f}  worst relative gradient error {worst:.2e}")
        assert not torch.equal(model.flat, before) and model.opt_step == 1


def test_training_batch_size_vs_the_cpu_oracle_and_loss_decreases():
    """A batch the fixture does not hold (4 samples of 96 x 96, one without second object) against oracle/train_oracle.py (given the
    engine's attention maps), then 8 more steps on the same batch: the loss must go down."""
    from oracle.make_golden_train import make_batch
    data = make_batch(dict(B=4, H=96, W=96, seed=5))
    data["dist"] = torch.tensor([[0.25, 0.75], [0.6, 0.4], [0.5, 0.5], [0.1, 0.9]])
    data["selector"] = torch.tensor([[1.0, 1.0], [1.0, 0.0], [1.0, 1.0], [1.0, 1.0]])
    model = _model(dict(lr=1e-3, steps=[1000], gamma=0.1, iterations=10))       # it = 3: inside the warm-up (top 57.5 %)
    out = model.do_pass(dict(data), 3)
    with torch.enable_grad():
        r = TO.train_step(Wt.make_fuse_state(0), data, out["attn1"].cpu(), out["attn2"].cpu(), 3, 10, 1e-3)
    assert abs(float(out["losses"]["total_loss"]) - r["total_loss"]) < 2e-4 * max(1.0, r["total_loss"])
    off = 0
    for n, p in model.net.named_parameters():
        ref_g = r["grads"][n]
        got_g = model.grad[off:off + p.numel()].view_as(p).cpu()
        assert float((got_g - ref_g).abs().max()) < 1e-3 * float(ref_g.abs().max()), n
        off += p.numel()
    first = float(out["losses"]["total_loss"])
    for step in range(8):
        out = model.do_pass(dict(data), 3)
    assert float(out["losses"]["total_loss"]) < first and model.opt_step == 9


class _Logger:
    def __init__(self):
        self.metrics, self.scalars = [], []

    def log_metrics(self, prefix, key, value, it, f=None):
        self.metrics.append((prefix, key, float(value), it))

    def log_scalar(self, tag, value, it):
        self.scalars.append((tag, float(value), it))


def test_checkpoint_round_trip_and_reference_layout(tmp_path):
    """save -> load_model resumes bit-identically; the file has the reference's layout (fusion_model.py:152-157: 'it', 'network',
    torch.optim.Adam's 'optimizer' state_dict, MultiStepLR's 'scheduler' state_dict) so that REAL torch objects - what the reference's
    FusionModel.load_model feeds (:171-173) - load it; a checkpoint written BY torch objects (the reference's side) resumes here; the
    periodic save inside do_pass (:123-125) and the integrator / finalize_val surface are there."""
    from oracle.make_golden_train import make_batch
    data = make_batch(dict(B=2, H=64, W=64, seed=77))
    para = dict(lr=1e-3, steps=[3], gamma=0.1, iterations=10)
    log = _Logger()
    a = _model(para, logger=log, save_path=str(tmp_path / "run" / "fusion"))
    a.save_model_interval, a.report_interval = 2, 2
    for it in range(4):
        a.do_pass(dict(data), it)                                   # it = 2: report + periodic checkpoint (it % interval == 0 and it != 0)
    assert os.path.isfile(str(tmp_path / "run" / "fusion_2.pth")) and os.path.isfile(str(tmp_path / "run" / "fusion_checkpoint.pth"))
    assert [m[:2] for m in log.metrics if m[3] == 2] == [("train", "time"), ("train", "total_loss"), ("train", "p"), ("train", "iou/iou"),
                                                          ("train", "iou/sec_iou")] and log.scalars[0][0] == "train/lr"      # fusion_model.py:38 + losses.py:8-16
    assert all(0.0 < m[2] <= 1.0 for m in log.metrics if m[1].startswith("iou/"))
    a.val().do_pass(dict(data), 4)
    a.finalize_val(4)
    assert ("val", "total_loss") in [m[:2] for m in log.metrics] and a.val_integrator.values == {}
    a.train()
    a.save_checkpoint(4)
    from collections import Counter
    with torch.serialization.safe_globals([Counter]):
        ck = torch.load(str(tmp_path / "run" / "fusion_checkpoint.pth"), map_location="cpu", weights_only=True)
    assert set(ck) == {"it", "network", "optimizer", "scheduler"} and ck["it"] == 4 and set(ck["optimizer"]) == {"state", "param_groups"}
    assert ck["scheduler"]["last_epoch"] == 4 and abs(ck["scheduler"]["_last_lr"][0] - 1e-4) < 1e-12          # milestone 3 passed
    # (1) real torch objects (the reference's load_model) accept the file
    ref_net = [torch.nn.Parameter(v.clone()) for k, v in ck["network"].items()]
    opt = torch.optim.Adam(ref_net, lr=1e-3, weight_decay=1e-7)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, [3], 0.1)
    opt.load_state_dict(ck["optimizer"])
    sch.load_state_dict(ck["scheduler"])
    assert abs(opt.param_groups[0]["lr"] - 1e-4) < 1e-12 and int(opt.state[ref_net[0]]["step"]) == 4
    # (2) round trip: a fresh model resumed from the file continues exactly like the original
    b = _model(para)
    assert b.load_model(str(tmp_path / "run" / "fusion_checkpoint.pth")) == 4 and b.opt_step == 4
    assert torch.equal(b.flat, a.flat) and torch.equal(b.exp_avg, a.exp_avg) and torch.equal(b.exp_avg_sq, a.exp_avg_sq)
    oa, ob = a.do_pass(dict(data), 5), b.do_pass(dict(data), 5)
    assert torch.equal(a.flat, b.flat) and float(oa["losses"]["total_loss"]) == float(ob["losses"]["total_loss"])
    # (3) a checkpoint written by torch objects (reference side): torch takes one more step on the engine's gradient, saves; the engine
    # loads it and must hold torch's state and then step like torch
    off = 0
    for p in ref_net:
        p.grad = a.grad[off:off + p.numel()].view_as(p).cpu().clone()
        off += p.numel()
    names = list(ck["network"])
    before = torch.cat([p.detach().reshape(-1) for p in ref_net])
    opt.step(); sch.step()
    torch.save({"it": 5, "network": {n: p.detach().clone() for n, p in zip(names, ref_net)}, "optimizer": opt.state_dict(), "scheduler": sch.state_dict()},
               str(tmp_path / "ref_checkpoint.pth"))
    c = _model(para)
    assert c.load_model(str(tmp_path / "ref_checkpoint.pth")) == 5 and c.opt_step == 5 and abs(c.current_lr() - 1e-4) < 1e-12
    assert torch.equal(c.flat.cpu(), torch.cat([p.detach().reshape(-1) for p in ref_net]))
    assert torch.equal(c.exp_avg.cpu(), torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in ref_net]))
    assert not torch.equal(before, c.flat.cpu())

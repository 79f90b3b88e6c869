"""Golden vectors of the FusionNet training step, produced by the UNMODIFIED reference modules on PyTorch-CPU.

TEST INFRASTRUCTURE ONLY (this container; `/root/reference` does not exist on the GPU box).  Writes
``tests/golden/train_small.npz``: for three iteration numbers (before / inside / after BootstrappedCE's warm-up) the inputs of
one `FusionModel.do_pass` call, and what the reference computes from them:

    reference code executed                                         stored
    model/attn_network.py AttentionReadNetwork.forward (no_grad)     attn1, attn2
    model/fusion_net.py FusionNet.forward x 2, torch.sigmoid         -
    model/aggregate.py aggregate_wbg_channel(prob * selector, True)  logits, mask
    model/losses.py LossComputer.compute (BootstrappedCE)            total_loss, p
    total_loss.backward()                                            grad.<parameter name>
    torch.optim.Adam(lr, weight_decay=1e-7).step()                   new.<parameter name>

`model/fusion_model.py:54-131` itself (DistributedDataParallel + .cuda()) cannot be constructed on CPU; the lines above are
its body, call for call (fusion_model.py:81-90, 92, 127-130).

    python -m oracle.make_golden_train
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_loader  # noqa: E402
from oracle import weights as Wt  # noqa: E402

CFG = dict(B=2, H=64, W=64, iterations=100, lr=1e-4, its=[0, 35, 60], seed=77)


def blobs(g, n, h, w, thr=0.0):
    """Smooth random binary masks [n,1,h,w]."""
    z = torch.randn(n, 1, h // 8, w // 8, generator=g)
    z = torch.nn.functional.interpolate(z, size=(h, w), mode="bilinear", align_corners=False)
    return (z > thr).float()


def make_batch(cfg):
    g = torch.Generator().manual_seed(cfg["seed"])
    B, H, W = cfg["B"], cfg["H"], cfg["W"]
    gt1, gt2 = blobs(g, B, H, W, 0.3), blobs(g, B, H, W, 0.5)
    gt2 = gt2 * (1 - gt1)
    soft = lambda m: (m * 0.8 + 0.1 + 0.1 * torch.randn(m.shape, generator=g)).clamp(0, 1)
    data = dict(rgb=torch.randn(B, 3, H, W, generator=g), src2_ref_im=torch.randn(B, 3, H, W, generator=g),
                gt=gt1, gt2=gt2, seg1=soft(gt1), seg2=soft(blobs(g, B, H, W, 0.3)), src2_ref=soft(gt1), src2_ref_gt=blobs(g, B, H, W, 0.3),
                seg12=soft(gt2), seg22=soft(blobs(g, B, H, W, 0.5)), src2_ref2=soft(gt2), src2_ref_gt2=blobs(g, B, H, W, 0.5),
                dist=torch.tensor([[0.25, 0.75], [0.6, 0.4]]), selector=torch.tensor([[1.0, 1.0], [1.0, 0.0]]))
    # the second sample has no second object (fusion_dataset.py:213-221)
    for k in ("gt2", "seg12", "seg22", "src2_ref2", "src2_ref_gt2"):
        data[k][1] = 0
    cls = torch.zeros(B, H, W, dtype=torch.long)
    cls[data["gt"][:, 0] > 0.5] = 1
    cls[data["gt2"][:, 0] > 0.5] = 2
    data["cls_gt"] = cls
    return data


def main():
    ref = ref_loader.load_reference()
    import importlib
    losses = importlib.import_module("model.losses")
    with contextlib.redirect_stdout(io.StringIO()):
        prop = ref["attn_network"].AttentionReadNetwork().eval()
    sd = Wt.make_prop_state(0)
    prop.load_state_dict({k: v for k, v in sd.items() if not k.startswith("decoder.")}, strict=False)
    data = make_batch(CFG)
    out = {"config": json.dumps(CFG)}
    for k, v in data.items():
        out["in." + k] = v.numpy()
    para = dict(iterations=CFG["iterations"])
    for it in CFG["its"]:
        net = ref["fusion_net"].FusionNet()
        net.load_state_dict(Wt.make_fuse_state(0))
        net.eval()                                                      # fusion_model.py:206-209 ("Also skip BN")
        opt = torch.optim.Adam(net.parameters(), lr=CFG["lr"], weight_decay=1e-7)
        with torch.no_grad():
            attn1, attn2 = prop(data["src2_ref_im"], data["src2_ref"], data["src2_ref_gt"], data["src2_ref2"], data["src2_ref_gt2"], data["rgb"])
        prob1 = torch.sigmoid(net(data["rgb"], data["seg1"], data["seg2"], attn1, data["dist"]))
        prob2 = torch.sigmoid(net(data["rgb"], data["seg12"], data["seg22"], attn2, data["dist"]))
        prob = torch.cat([prob1, prob2], 1) * data["selector"].unsqueeze(2).unsqueeze(2)
        logits, mask = ref["aggregate"].aggregate_wbg_channel(prob, True)
        lc = losses.LossComputer(para)
        ls = lc.compute({**data, "logits": logits, "mask": mask}, it)
        opt.zero_grad(set_to_none=True)
        ls["total_loss"].backward()
        tag = f"it{it}."
        out[tag + "attn1"], out[tag + "attn2"] = attn1.numpy(), attn2.numpy()
        out[tag + "logits"], out[tag + "mask"] = logits.detach().numpy(), mask.detach().numpy()
        out[tag + "total_loss"], out[tag + "p"] = np.float64(ls["total_loss"].item()), np.float64(float(ls["p"]))
        for n, p in net.named_parameters():
            out[tag + "grad." + n] = p.grad.numpy().copy()
        opt.step()
        for n, p in net.named_parameters():
            out[tag + "new." + n] = p.detach().numpy().copy()
        print(f"it {it}: total_loss {ls['total_loss'].item():.6f}  p {float(ls['p']):.4f}  |grad| max {max(float(p.grad.abs().max()) for p in net.parameters()):.3e}")
    path = os.path.join(ROOT, "tests", "golden", "train_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
